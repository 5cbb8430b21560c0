#!/usr/bin/env python3
"""Experiment for a streamed WRITER: how many rows of chunk k are still undecided when the chunk is finalized
right after its own search, with hmmsearch's domZ known only as [counts of chunks <= k, those counts' upper bounds + the reads not
yet seen]?  Rows decided under these wider bounds stay decided under the final ones (the interval only shrinks), so every read
that depends on decided rows alone could be trimmed and deflated while the GPU still works on later chunks.
usage: stream_provisional.py [--reads 3000000] [--chunk-mb 0]"""
import argparse
import gzip
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=3000000)
    ap.add_argument("--chunk-mb", type=float, default=0.0)
    args = ap.parse_args()
    import synth
    from bench import its2_profiles
    from itsxpress_amd.stream import StreamEngine
    from itsxpress_amd.trim import write_trimmed_fastq
    with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f:
        thmm = f.read()
    blob, offs = synth.make_reads(thmm, args.reads, config=3, seed=synth.SEED + 3, fixed_len=0, len_range=(300, 580))
    n = args.reads
    tmp = tempfile.mkdtemp(prefix="itsx_prov_")
    try:
        plain = os.path.join(tmp, "in.fastq")
        bases = np.frombuffer(blob, np.uint8)
        with open(plain, "wb") as f:
            q = b"I" * 600
            for i in range(n):
                s = bases[offs[i]:offs[i + 1]]
                f.write(b"@read%d\n" % i + s.tobytes() + b"\n+\n" + q[:len(s)] + b"\n")
        fq = os.path.join(tmp, "in.fastq.gz")
        write_trimmed_fastq(plain, fq, np.zeros(n, np.int32), np.full(n, 1 << 30, np.int32), gzipped=True)
        os.remove(plain)
        se = StreamEngine(0, chunk_mb=args.chunk_mb or None)
        se.set_rows_mode("lazy")
        se.load_reads_file(fq)
        se.derep()
        se.load_profiles(text=its2_profiles(thmm))
        se.search()
        zs = [np.asarray(z, np.int64) for z in se._z]
        half = zs[0].shape[0] // 2
        seen = 0
        rows = []
        acc = np.zeros_like(zs[0])
        for k, ((eng, st), z) in enumerate(zip(se._engs, zs)):
            acc = acc + z
            seen += eng.n_reads
            prov = acc.copy()
            prov[half:] += n - seen                    # every read not yet seen may add one reported target to every profile
            t0 = time.perf_counter()
            eng.set_domz(prov)
            eng.finalize(domE=10.0)
            rows.append({"chunk": k, "reads": eng.n_reads, "uniques": eng.n_unique, "future_reads": n - seen,
                         "undecided_rows_that_matter": eng.lazy_pending(), "profiles_flagged": int(eng.lazy_pending_profiles().sum()),
                         "finalize_ms": round((time.perf_counter() - t0) * 1e3, 1)})
        se.finalize()                                  # the real thing, with the summed counters
        final = {"undecided_after_final_bounds": [int(e.lazy_pending()) for e, _ in se._engs]}
        print(json.dumps({"reads": n, "chunks": se.world, "zlb_sum": int(acc[:half].sum()), "zub_sum": int(acc[half:].sum()),
                          "provisional": rows, **final}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
