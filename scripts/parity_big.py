#!/usr/bin/env python3
"""One-off larger equality check of the HMM stages: N synthetic reads (default 30 000, ~20 k uniques) x the 155 ITS2
profiles of the stand-in taxon, engine against the CPU oracle on every compared quantity of tests/test_gpu_parity.py
(filter trace bits, domain rows, score bits, coordinates).  usage: parity_big.py [reads]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
thmm = gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt").read()
hmm = tp._its2_subset(thmm)
blob, offs = synth.make_reads(thmm, n, config=2, seed=synth.SEED + 77)
seqs = synth.to_strings(blob, offs)
rng = np.random.default_rng(5)
seqs += [s[:int(rng.integers(120, 299))] for s in seqs[:2000]]          # ragged lengths as well
eng = Engine(0)
t0 = time.time()
res = tp._run_both(eng, hmm, seqs, threads=os.cpu_count() or 8)
print("ran both in %.1f s: %d uniques, %d pairs past MSV, %d domains" % (time.time() - t0, eng.n_unique, res.counts["past_msv"], len(res.domains)), flush=True)
tp._compare(eng, res)
print("engine == oracle on every compared quantity")
