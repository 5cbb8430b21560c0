#!/usr/bin/env python3
"""VALU issue rates on this MI355X (round 5, the review's item 3a): cycles per wave instruction for the instruction classes the two
scan kernels are made of, with no dependence between consecutive instructions, at 1 / 2 / 4 / 6 / 8 waves per SIMD on every CU.
Prints a markdown table (profiles/round5_valu_issue.md) and a JSON line with the SIMD's issue interval per class at full occupancy --
what bench.py prices valu_issue_frac with.   usage: python scripts/valu_issue.py [--iters 20000]"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itsxpress_amd import Engine

OPS = ["v_fma_f32", "v_pk_fma_f32", "v_pk_max_i16", "v_pk_add_u16", "v_pk_mul_f32", "v_pk_add_f32", "s_nop 0", "v_mul_f32", "v_pk_mov_b32", "v_max_i16"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20000)
    a = ap.parse_args()
    eng = Engine(0)
    waves = [1, 2, 4, 6, 8]
    rows = {}
    for op, name in enumerate(OPS):
        rows[name] = []
        for w in waves:
            cyc, ms = C.c_double(), C.c_double()
            eng._chk(eng.L.itsx_debug_issue(eng.h, op, w, a.iters, C.byref(cyc), C.byref(ms)))
            # cyc.value: ticks per instruction as one wave sees them; the SIMD issues one instruction of this class every cyc / w ticks
            rows[name].append((cyc.value, cyc.value / w, ms.value))
            # the launch's wall time against the waves' own ticks: co-residency check (a launch that ran in two rounds takes twice its waves' time)
            rows[name][-1] += (ms.value * 1e-3 / (cyc.value * 64.0 * a.iters),)          # seconds per tick, if every wave ran all the time
    print("| instruction | " + " | ".join("%d wave%s / SIMD: per wave, per SIMD" % (w, "" if w == 1 else "s") for w in waves) + " |")
    print("|---|" + "---|" * len(waves))
    for name in OPS:
        print("| `%s` | " % name + " | ".join("%.2f, %.2f" % (r[0], r[1]) for r in rows[name]) + " |")
    print()
    print("implied clock (GHz) = ticks of one wave / the launch's wall time; a launch whose blocks were not all resident at once reads about half:")
    print()
    print("| instruction | " + " | ".join("%d" % w for w in waves) + " |")
    print("|---|" + "---|" * len(waves))
    for name in OPS:
        print("| `%s` | " % name + " | ".join("%.2f" % (1e-9 / r[3]) for r in rows[name]) + " |")
    print()
    print(json.dumps({"issue_cycles_per_simd": {name: round(min(r[1] for r in rows[name]), 3) for name in OPS}, "iters": a.iters,
                      "ms": {name: [round(r[2], 3) for r in rows[name]] for name in OPS}}))


if __name__ == "__main__":
    main()
