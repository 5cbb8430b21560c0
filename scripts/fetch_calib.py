#!/usr/bin/env python3
"""Streams of known size in the DP slab's access patterns, for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/calib_fetch -o c -- python3 scripts/fetch_calib.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/calib_write -o c -- python3 scripts/fetch_calib.py

Prints, per pattern, the bytes one launch touches; scripts/pmc_summary.py divides the counter by it.
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from itsxpress_amd import Engine

eng = Engine(0)
out = {}
for pattern, name in ((0, "k_calib_read4 (6 of 6 fields, 4 B/lane)"), (1, "k_calib_read4 (5 of 6 fields)"), (2, "k_calib_write4"),
                      (3, "k_calib_read16")):
    b, ms = C.c_int64(0), C.c_double(0)
    eng._chk(eng.L.itsx_debug_calibrate(eng.h, pattern, 4.0, 3, C.byref(b), C.byref(ms)))
    out[name] = {"pattern": pattern, "bytes_per_launch": b.value, "ms_per_launch": ms.value, "GBps": b.value / ms.value / 1e6}
print(json.dumps(out))
