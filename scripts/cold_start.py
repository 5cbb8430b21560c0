#!/usr/bin/env python3
"""What a one-shot run pays: a FRESH process per measurement, N synthetic 300-bp reads through the whole path once (first
search of the context, device memory allocated on the way), then a second time (steady state).
usage: cold_start.py [reads ...]   (ITSX_SLAB_ADAPT=0 in the environment shows the fixed 64-GB slab budget)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time, gzip
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import synth
from bench import its2_profiles
from itsxpress_amd import Engine
n = int(sys.argv[1])
thmm = gzip.open(%r + "/tests/golden/T.hmm.gz", "rt").read()
blob, offs = synth.make_reads(thmm, n, config=2, seed=synth.SEED + 2)
t0 = time.perf_counter()
eng = Engine(0); eng.load_profiles(text=its2_profiles(thmm)); eng.set_reads_buffer(blob, offs)
t1 = time.perf_counter()
out = []
for rep in range(3):
    a = time.perf_counter(); eng.derep(); eng.search(); eng.finalize(); eng.trim_coords("3_", "4_"); out.append(time.perf_counter() - a)
print("%%d reads: context+profiles+hand-over %%.2f s, first pass %%.2f s, second %%.2f s, third %%.2f s  (%%d batches)" %% (n, t1 - t0, out[0], out[1], out[2], eng.stats()["n_batches"]), flush=True)
''' % (ROOT, ROOT, ROOT)

for n in (sys.argv[1:] or ["50000", "200000", "1000000"]):
    for adapt in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", CHILD, n], env=dict(os.environ, ITSX_SLAB_ADAPT=adapt), capture_output=True, text=True)
        print("adaptive slab" if adapt == "1" else "fixed budget ", (r.stdout.strip() or r.stderr[-300:]), flush=True)
