#!/usr/bin/env python3
"""Scale check on one GPU: N synthetic 300-bp reads generated in 1 M-read pieces (bounded host memory), then the
whole path once.  Prints one JSON line (reads/s, uniques, chunks, stage times) and checks the size-independent
properties: every read of a cluster carries its representative's coordinates, dropped reads are -1.
usage: scale_run.py [--reads 10000000] [--cluster-id 1.0]"""
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
    ap.add_argument("--reads", type=int, default=10000000)
    ap.add_argument("--cluster-id", type=float, default=1.0)
    ap.add_argument("--passes", type=int, default=1, help="run the path this many times in the same context and report the last (2 = steady state)")
    args = ap.parse_args()
    import synth
    from bench import its2_profiles
    from itsxpress_amd import Engine
    with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f:
        thmm = f.read()
    piece = 1000000
    blobs, lens = [], []
    t0 = time.perf_counter()
    for k in range(0, args.reads, piece):
        n = min(piece, args.reads - k)
        # the same template library in every piece (seeded), fresh errors: duplicates across pieces like a real run
        b, o = synth.make_reads(thmm, n, config=2, seed=synth.SEED + 2, frac_templates=0.02 * piece / max(n, 1))
        rng = np.random.default_rng(1000 + k)
        arr = np.frombuffer(b, np.uint8).copy()
        flip = rng.random(arr.shape[0]) < 0.002
        arr[flip] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(flip.sum()))]
        blobs.append(arr.tobytes())
        lens.append(np.diff(o))
    blob = b"".join(blobs)
    offs = np.zeros(args.reads + 1, np.int64)
    np.cumsum(np.concatenate(lens), out=offs[1:])
    t_gen = time.perf_counter() - t0
    eng = Engine(0)
    nprof = eng.load_profiles(text=its2_profiles(thmm))
    t0 = time.perf_counter()
    eng.set_reads_buffer(blob, offs)
    t_pack = time.perf_counter() - t0
    times = []
    for _ in range(max(1, args.passes)):
        t0 = time.perf_counter()
        if args.cluster_id < 1.0:
            eng.cluster(args.cluster_id)
        else:
            eng.derep()
        eng.search()
        eng.finalize()
        start, stop, tlen, ind = eng.trim_coords("3_", "4_")
        times.append(time.perf_counter() - t0)
    dt = times[-1]
    rep_of, strand, uniq_of = eng.get_derep()
    us, ue, ut, ui = eng.rep_coords("3_", "4_")
    ok = uniq_of >= 0
    assert np.array_equal(start[ok], us[uniq_of[ok]]) and np.array_equal(stop[ok], ue[uniq_of[ok]])
    assert (start[~ok] == -1).all() and np.array_equal(rep_of[rep_of[ok]], rep_of[ok])
    st = eng.stats()
    print(json.dumps({"reads": args.reads, "profiles": nprof, "reads_per_s": args.reads / dt, "seconds": dt, "seconds_per_pass": [round(x, 2) for x in times], "gen_s": t_gen,
                      "pack_upload_s": t_pack, "unique": st["n_unique"], "pairs_past_msv": st["n_past_msv"],
                      "domains": st["n_domains"], "trimmed": int(((start >= 0) & (stop > start)).sum()),
                      "domain_overflow": st["n_domain_overflow"],
                      "stage_ms": {k: round(v, 1) for k, v in st.items() if k.startswith("ms_")}}))


if __name__ == "__main__":
    main()
