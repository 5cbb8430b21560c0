import sys, os, time, gzip, json
import numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+"/tests")
import synth
from bench import its2_profiles
from itsxpress_amd import Engine
thmm=gzip.open(ROOT+"/tests/golden/T.hmm.gz","rt").read()
eng=Engine(0); eng.load_profiles(text=its2_profiles(thmm))
for n in (227, 2000, 10000, 50000, 100000):
    blob,offs=synth.make_reads(thmm,n,config=2,seed=synth.SEED+2)
    ts=[]
    for rep in range(4):
        t0=time.perf_counter(); eng.set_reads_buffer(blob,offs); t1=time.perf_counter()
        eng.derep(); t2=time.perf_counter(); eng.search(); t3=time.perf_counter(); eng.finalize(); t4=time.perf_counter()
        eng.trim_coords("3_","4_"); t5=time.perf_counter()
        ts.append((t1-t0,t2-t1,t3-t2,t4-t3,t5-t4))
    print(n, "set %.1f derep %.1f search %.1f finalize %.1f coords %.1f ms (4th rep) | first rep total %.1f ms"%(tuple(1e3*x for x in ts[-1])+(1e3*sum(ts[0]),)))
