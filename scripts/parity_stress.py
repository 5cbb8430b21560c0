#!/usr/bin/env python3
"""Equality check of the HMM stages on reads built to stress the per-lane paths of the kernels: many degenerate bases (2 % N, the
exception lists of k_msv / the DP kernels), every length from 20 to 620 mixed inside the waves, a tenth of the reads random (fail
the filters at different rows), engine against the CPU oracle on every compared quantity of tests/test_gpu_parity.py.
usage: parity_stress.py [reads] [seed]"""
import gzip
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import synth  # noqa: E402
import test_gpu_parity as tp  # noqa: E402
from itsxpress_amd import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
thmm = gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt").read()
hmm = tp._its2_subset(thmm)
blob, offs = synth.make_reads(thmm, n, config=3, seed=synth.SEED + seed, fixed_len=0, len_range=(300, 580), n_rate=0.02, sub_rate=0.01)
seqs = synth.to_strings(blob, offs)
rng = np.random.default_rng(seed)
out = []
for i, s in enumerate(seqs):
    r = rng.random()
    if r < 0.35:                                    # cut to any length, from either end
        k = int(rng.integers(20, len(s)))
        s = s[:k] if rng.random() < 0.5 else s[-k:]
    elif r < 0.45:                                  # random sequence with degenerate codes
        s = "".join(rng.choice(list("ACGTNRYKMSWBDHV"), size=int(rng.integers(20, 620))))
    elif r < 0.50:                                  # longer than any template: the motif twice
        s = s + s[: int(rng.integers(10, 200))]
    out.append(s)
eng = Engine(0)
t0 = time.time()
res = tp._run_both(eng, hmm, out, threads=os.cpu_count() or 8)
print("ran both in %.1f s: %d uniques, %d pairs past MSV, %d domains" % (time.time() - t0, eng.n_unique, res.counts["past_msv"], len(res.domains)), flush=True)
tp._compare(eng, res)
print("engine == oracle on every compared quantity")
