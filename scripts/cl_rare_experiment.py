#!/usr/bin/env python3
"""configs[4], the review's item 6: can the O(queries x centroids) word counting of `--cluster_size` be pruned by RARE words?
(CPU only, the oracle's clustering; ~3 min at the default size.)

vsearch ranks every centroid by the 8-mers it shares with the query and tries the 32 best (itsxpress/SeqSample.py:147-161); the engine
streams every centroid past every query (k_cl_stream: DESIGN 4a).  Proposal: split the query's words into rare ones (held by fewer than
T centroids: short posting lists) and conserved ones.  count(c) <= rare_hits(c) + n_conserved(q), so only the centroids whose BOUND
reaches the 32nd best exact count need the conserved words' bitmaps looked at.  This script measures, on configs[4]'s own generator,
how large that candidate set is: for a sample of query strands the exact counts against every centroid (a sparse matrix product), the
32nd best, and the number of centroids with rare_hits + n_conserved >= it -- for several T.  Build only if it is < 5 % of the centroids.

usage: cl_rare_experiment.py [--reads 30000] [--sample 400]   -> one JSON line per T (profiles/round5_cluster_rare_words.jsonl)"""
import argparse
import gzip
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def words_of(seq_codes):
    """distinct unambiguous 8-mers (16-bit codes) of one strand given as an array of 0..3 (4 = ambiguous)"""
    n = len(seq_codes)
    if n < 8:
        return np.zeros(0, np.int64)
    c = seq_codes.astype(np.int64)
    w = np.zeros(n - 7, np.int64)
    bad = np.zeros(n - 7, bool)
    for k in range(8):
        w = (w << 2) | (c[k:n - 7 + k] & 3)
        bad |= c[k:n - 7 + k] > 3
    return np.unique(w[~bad])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=30000)
    ap.add_argument("--sample", type=int, default=400)
    args = ap.parse_args()
    import orc
    import synth
    with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f:
        thmm = f.read()
    # configs[4]'s generator as bench.py --workload cfg4 calls it
    blob, offs = synth.make_reads(thmm, args.reads, config=5, left="1_", right="4_", fixed_len=0, len_range=(300, 480))
    seqs = synth.to_strings(blob, offs)
    t0 = time.time()
    codes, o = orc.digitize(seqs)
    res = orc.cluster(codes, o, cluster_id=0.995, strand_both=True)
    rep = np.asarray(res["rep_of"])
    cent = np.flatnonzero(rep == np.arange(len(seqs)))
    print("clustered %d reads -> %d centroids in %.0f s" % (len(seqs), len(cent), time.time() - t0), file=sys.stderr, flush=True)
    lut = np.full(256, 4, np.int64)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    comp = np.array([3, 2, 1, 0, 4])

    def strands(s):
        c = lut[np.frombuffer(s.encode(), np.uint8)]
        return c, comp[c][::-1]
    # centroid x word incidence (forward strand of the centroids: what the index holds)
    rows, cols = [], []
    for j, r in enumerate(cent):
        w = words_of(strands(seqs[r])[0])
        rows.append(np.full(len(w), j, np.int64)); cols.append(w)
    A = sp.csr_matrix((np.ones(sum(len(x) for x in cols), np.int32), (np.concatenate(rows), np.concatenate(cols))), shape=(len(cent), 65536))
    post = np.asarray(A.sum(axis=0)).ravel()                 # posting-list length of every word
    Ac = A.tocsc()
    rng = np.random.default_rng(3)
    qs = rng.choice(len(seqs), size=min(args.sample, len(seqs)), replace=False)
    out = {}
    for T in (8, 32, 128, 512):
        out[T] = {"cand_frac": [], "n_cons": [], "t32": [], "kind": []}
    for r in qs:
        before = int(np.searchsorted(cent, r))               # the query sees the centroids made before it (label order = input order here)
        if before < 64:
            continue
        for strand_no, c in enumerate(strands(seqs[r])):
            w = words_of(c)
            if len(w) == 0:
                continue
            sub = Ac[:before, :][:, w] if False else Ac[:, w]
            cnt_all = np.asarray(sub.sum(axis=1)).ravel()[:before]
            t32 = np.sort(cnt_all)[-32] if before >= 32 else 0
            for T in out:
                rare = post[w] < T
                n_cons = int((~rare).sum())
                rh = np.asarray(Ac[:, w[rare]].sum(axis=1)).ravel()[:before] if rare.any() else np.zeros(before, np.int64)
                cand = int(((rh + n_cons) >= max(t32, 1)).sum())
                out[T]["cand_frac"].append(cand / before); out[T]["n_cons"].append(n_cons); out[T]["t32"].append(int(t32)); out[T]["kind"].append(strand_no)
    for T, d in out.items():
        cf = np.array(d["cand_frac"])
        hist, edges = np.histogram(cf, bins=[0, 0.001, 0.01, 0.05, 0.2, 0.5, 1.0001])
        print(json.dumps({"reads": len(seqs), "centroids": int(len(cent)), "rare_below": T, "query_strands": int(len(cf)),
                          "candidate_fraction_mean": round(float(cf.mean()), 4), "median": round(float(np.median(cf)), 4),
                          "p10": round(float(np.quantile(cf, 0.1)), 4), "p90": round(float(np.quantile(cf, 0.9)), 4),
                          "histogram(<0.1%,<1%,<5%,<20%,<50%,<=100%)": hist.tolist(),
                          "conserved_words_per_query_mean": round(float(np.mean(d["n_cons"])), 1), "t32_mean": round(float(np.mean(d["t32"])), 1),
                          "plus_strand_mean": round(float(cf[np.array(d["kind"]) == 0].mean()), 4), "minus_strand_mean": round(float(cf[np.array(d["kind"]) == 1].mean()), 4)}), flush=True)


if __name__ == "__main__":
    main()
