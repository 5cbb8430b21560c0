#!/usr/bin/env python3
"""f2 measurement: N synthetic 2x250 pairs (amplicons of 300-480 bases, Illumina-like qualities) through the merge
kernel; prints pairs/s from the kernel's HIP-event time (inputs resident in HBM), the end-to-end rate of the buffer
API (upload + kernel + download), and the CPU oracle's rate on a sample.  usage: merge_bench.py [--pairs 1000000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_pairs(n, seed=3):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    L = rng.integers(300, 481, n)
    frag = acgt[rng.integers(0, 4, (n, 480))]
    qv = np.array([2, 12, 22, 30, 37, 38], np.uint8)
    qp = [.001, .009, .03, .08, .28, .6]

    def side(reads):
        q = rng.choice(qv, size=reads.shape, p=qp)
        err = rng.random(reads.shape) < 10.0 ** (-q.astype(np.float64) / 10.0)
        reads = reads.copy()
        reads[err] = acgt[rng.integers(0, 4, int(err.sum()))]
        return reads, (q + 33).astype(np.uint8)
    fwd = frag[:, :250]
    rev = np.empty((n, 250), np.uint8)
    for i in range(n):                                   # reverse read = reverse complement of the fragment's last 250 bases
        rev[i] = comp[frag[i, L[i] - 250:L[i]][::-1]]
    fwd, fq = side(fwd)
    rev, rq = side(rev)
    return fwd, fq, rev, rq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1000000)
    ap.add_argument("--cpu-sample", type=int, default=2000)
    args = ap.parse_args()
    import ctypes as C
    import orc
    from itsxpress_amd import Engine
    n = args.pairs
    fwd, fq, rev, rq = make_pairs(n)
    off = np.arange(n + 1, dtype=np.int64) * 250
    eng = Engine(0)
    cap = 500 * n + 1
    oseq, oqual = C.create_string_buffer(cap), C.create_string_buffer(cap)
    olen, reason = np.zeros(n, np.int32), np.zeros(n, np.int32)
    bufs = [x.tobytes() for x in (fwd, fq, rev, rq)]
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        eng._chk(eng.L.itsx_merge_buffers(eng.h, bufs[0], bufs[1], off.ctypes.data, bufs[2], bufs[3], off.ctypes.data, n, 40, 2.0, 0,
                                          oseq, oqual, olen.ctypes.data, reason.ctypes.data, None, None))
        dt = time.perf_counter() - t0
        ms = eng.stats()["ms_merge"]
        best = (ms, dt) if best is None or ms < best[0] else best
    m = min(args.cpu_sample, n)
    sample = [(fwd[i].tobytes().decode(), fq[i].tobytes().decode(), rev[i].tobytes().decode(), rq[i].tobytes().decode()) for i in range(m)]
    t0 = time.perf_counter()
    res = [orc.merge_pair(*p) for p in sample]
    cdt = time.perf_counter() - t0
    for i, r in enumerate(res):
        assert (r[0] == "ok") == (reason[i] == 0) and (r[0] != "ok" or oseq[500 * i:500 * i + int(olen[i])].decode() == r[1])
    print(json.dumps({"pairs": n, "merged": int((reason == 0).sum()), "kernel_ms": best[0], "pairs_per_s_kernel": n / (best[0] * 1e-3),
                      "pairs_per_s_buffer_api": n / best[1], "bytes_per_pair_in": 1000, "kernel_GBps_in": n * 1000 / (best[0] * 1e-3) / 1e9,
                      "cpu_oracle_pairs_per_s_1_thread": m / cdt, "cpu_sample": m, "sample_agrees": True}))


if __name__ == "__main__":
    main()
