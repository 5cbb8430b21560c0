#!/usr/bin/env python3
"""One-off larger equality check for row a2: N bench-like synthetic reads clustered by the engine and by the sequential CPU
oracle (minutes of CPU for 20 k reads); prints whether every outcome agrees.  usage: cluster_check.py [--reads 20000] [--id 0.995]"""
import argparse
import gzip
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--id", type=float, default=0.995)
    args = ap.parse_args()
    import orc
    import synth
    from itsxpress_amd import Engine
    with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f:
        thmm = f.read()
    blob, offs = synth.make_reads(thmm, args.reads, config=2, seed=synth.SEED + 77)
    reads = synth.to_strings(blob, offs)
    rng = np.random.default_rng(5)
    names = ["M%07d" % int(x) for x in rng.permutation(10 ** 6)[:len(reads)]]
    eng = Engine(0)
    eng.set_reads(reads, names)
    t0 = time.perf_counter()
    eng.cluster(args.id)
    t_gpu = time.perf_counter() - t0
    rep_of, strand, _ = eng.get_derep()
    pct, order = eng.get_cluster()
    st = eng.stats()
    codes, off = orc.digitize(reads)
    t0 = time.perf_counter()
    o = orc.cluster(codes, off, names, args.id)
    t_cpu = time.perf_counter() - t0
    same = bool(np.array_equal(order, o["order"]) and np.array_equal(rep_of, o["rep_of"]) and np.array_equal(strand, o["strand"])
                and np.array_equal(pct.view(np.uint64), o["pct_id"].view(np.uint64)))
    print(json.dumps({"reads": args.reads, "id": args.id, "identical": same, "centroids": o["n_centroids"], "engine_s": t_gpu, "oracle_s": t_cpu,
                      "oracle_alignments": o["n_alignments"], "engine_full_alignments": st["cl_alignments"], "engine_certified": st["cl_certified"],
                      "windows": st["cl_windows"], "cut_windows": st["cl_cuts"]}))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
