#!/usr/bin/env python3
"""VALU issue rates on this MI355X (round 5, the review's item 3a): what ONE instruction of each class costs a SIMD when the two scan
kernels' instruction classes are issued back to back with no dependence between consecutive instructions (16 accumulators in rotation),
at 1 / 2 / 4 / 6 / 8 waves per SIMD on every CU (itsx_debug_issue, csrc/k_util.hip: k_issue).

Two readings per cell: the launch's WALL time over the instructions one SIMD issued (ns per instruction per SIMD: independent of how the
blocks were placed and of the clock the chip held), and s_memtime ticks per instruction as one wave sees them.  Prints a markdown table
(profiles/round5_valu_issue.md) and a JSON line (profiles/round5_valu_issue.json: what bench.py prices valu_issue_frac with).
usage: python scripts/valu_issue.py [--iters 20000]"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itsxpress_amd import Engine

OPS = ["v_fma_f32", "v_pk_fma_f32", "v_pk_max_i16", "v_pk_add_u16", "v_pk_mul_f32", "v_pk_add_f32", "s_nop 0", "v_mul_f32", "v_pk_mov_b32", "v_max_i16"]
PACKED = ["v_pk_fma_f32", "v_pk_max_i16", "v_pk_add_u16", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_mov_b32"]
PLAIN = ["v_mul_f32", "v_max_i16"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20000)
    a = ap.parse_args()
    eng = Engine(0)
    waves = [1, 2, 4, 6, 8]
    ns, ticks = {}, {}
    for op, name in enumerate(OPS):
        ns[name], ticks[name] = [], []
        for w in waves:
            cyc, ms = C.c_double(), C.c_double()
            eng._chk(eng.L.itsx_debug_issue(eng.h, op, w, a.iters, C.byref(cyc), C.byref(ms)))
            ns[name].append(ms.value * 1e6 / (64.0 * a.iters * w))          # the SIMD issued 64 x iters x w instructions in ms
            ticks[name].append(cyc.value)
    print("ns per instruction per SIMD (launch wall time / instructions a SIMD issued):\n")
    print("| instruction | " + " | ".join("%d wave%s / SIMD" % (w, "" if w == 1 else "s") for w in waves) + " |")
    print("|---|" + "---|" * len(waves))
    for name in OPS:
        print("| `%s` | " % name + " | ".join("%.2f" % v for v in ns[name]) + " |")
    print("\ns_memtime ticks per instruction as ONE wave sees them (a wave alone on its SIMD issues one instruction per 4.4-5.3 ticks):\n")
    print("| instruction | " + " | ".join("%d" % w for w in waves) + " |")
    print("|---|" + "---|" * len(waves))
    for name in OPS:
        print("| `%s` | " % name + " | ".join("%.2f" % v for v in ticks[name]) + " |")
    print()
    sat = lambda names: sum(min(ns[n][2:]) for n in names) / len(names)
    print(json.dumps({"ns_per_simd": {"packed": round(sat(PACKED), 3), "plain": round(sat(PLAIN), 3), "s_nop": round(min(ns["s_nop 0"][2:]), 3),
                                      "v_fma_f32": round(min(ns["v_fma_f32"][2:]), 3)},
                      "note": "saturated (the smallest of the 4 / 6 / 8-wave cells) wall time of scripts/valu_issue.py's launches over the instructions a SIMD issued; "
                              "packed = mean of v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32, v_pk_max_i16, v_pk_add_u16, v_pk_mov_b32; plain = mean of v_mul_f32, v_max_i16",
                      "iters": a.iters}))


if __name__ == "__main__":
    main()
