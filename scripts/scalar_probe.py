#!/usr/bin/env python3
"""What the scalar cache gives a CU (round 5): k_fwd_bound takes its profile's transitions as 64 B per pair of nodes through two
s_load_dwordx8, waited for one step later.  Op 10 = those loads alone, op 11 = the loads under 12 independent v_pk_fma_f32 per step (the
kernel's own ratio), op 1 = the packed instructions alone -- at 1 / 2 / 3 / 4 waves per SIMD on every CU (itsx_debug_issue).
Prints ns per STEP per SIMD (launch wall time over the steps one SIMD issued) and the scalar bytes per ns per CU that is.
usage: python scripts/scalar_probe.py [--iters 4000]"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itsxpress_amd import Engine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4000)
    a = ap.parse_args()
    eng = Engine(0)
    out = {}
    for op, name, per_step in ((10, "loads alone", 1), (11, "loads + 12 v_pk_fma_f32", 1), (1, "12 v_pk_fma_f32 alone", 12)):
        row = []
        for w in (1, 2, 3, 4):
            cyc, ms = C.c_double(), C.c_double()
            eng._chk(eng.L.itsx_debug_issue(eng.h, op, w, a.iters, C.byref(cyc), C.byref(ms)))
            ns_step = ms.value * 1e6 / (64.0 * a.iters * w) * per_step       # a SIMD issued 64 x iters x w steps (op 1: instructions)
            row.append({"waves_per_simd": w, "ns_per_step_per_simd": round(ns_step, 2),
                        "scalar_B_per_ns_per_CU": None if op == 1 else round(4 * 64.0 / ns_step, 1)})
        out[name] = row
        print(name, json.dumps(row))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
