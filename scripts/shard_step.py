#!/usr/bin/env python3
"""Where a shard-sized step spends its time, call by call (derep / search / finalize / coordinates) against the engine's own stage timers:
what is left over is host work between the kernels.  usage: shard_step.py [reads]"""
import sys, os, time, gzip, json
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import numpy as np, synth
from bench import its2_profiles
from itsxpress_amd import Engine
thmm = gzip.open(ROOT+"/tests/golden/T.hmm.gz","rt").read()
n=int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
blob, offs = synth.make_reads(thmm, n, config=3, seed=synth.SEED+3, fixed_len=0, len_range=(300,580))
e=Engine(0); e.set_rows_mode("lazy"); e.load_profiles(text=its2_profiles(thmm))
e.set_reads_buffer(blob, offs)
for it in range(4):
    t0=time.perf_counter(); e.derep(); t1=time.perf_counter(); e.search(); t2=time.perf_counter(); e.finalize(); t3=time.perf_counter(); c=e.trim_coords("3_","4_"); t4=time.perf_counter()
    st=e.stats()
    print("derep %.1f search %.1f finalize %.1f coords %.1f total %.1f | stage sum msv %.1f filters %.1f domains %.1f" % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3,(t4-t0)*1e3, st["ms_msv"], st["ms_filters"], st["ms_domains"]))
