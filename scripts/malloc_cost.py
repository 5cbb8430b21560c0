#!/usr/bin/env python3
"""hipMalloc cost against size, each size in a fresh process (and the same size twice in one process)."""
import subprocess
import sys
CHILD = r'''
import ctypes, sys, time
hip = ctypes.CDLL("libamdhip64.so")
gb = float(sys.argv[1])
p = ctypes.c_void_p()
hip.hipSetDevice(0)
t0 = time.perf_counter(); hip.hipFree(None); t1 = time.perf_counter()
n = ctypes.c_size_t(int(gb * (1 << 30)))
rc = hip.hipMalloc(ctypes.byref(p), n); t2 = time.perf_counter()
hip.hipFree(p); t3 = time.perf_counter()
rc2 = hip.hipMalloc(ctypes.byref(p), n); t4 = time.perf_counter()
hip.hipMemset(p, 0, n); hip.hipDeviceSynchronize(); t5 = time.perf_counter()
print("%5.1f GB: init %.2f s, hipMalloc %.3f s (rc %d), hipFree %.3f s, again %.3f s, memset %.3f s" % (gb, t1 - t0, t2 - t1, rc, t3 - t2, t4 - t3, t5 - t4), flush=True)
'''
for gb in (sys.argv[1:] or ["1", "4", "8", "16", "24", "32", "48", "64", "72", "4", "64"]):
    r = subprocess.run([sys.executable, "-c", CHILD, gb], capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-300:], flush=True)
