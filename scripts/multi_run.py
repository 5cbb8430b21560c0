#!/usr/bin/env python3
"""ITSXPRESS_GPUS=N behind the mirror, at size (round 5, the review's item 2): ONE sample of --reads synthetic merged reads through
`MultiEngine` with N worker processes time-sliced on ONE GPU (a GPU box has one; at most 6 processes may share it), in arrays mode --
load (parent inflates once, cuts record-aligned pieces), exact global dereplication (hash-partitioned owner step IN the workers),
search, finalize, per-read coordinates (composed IN the workers) -- and what THIS process spent per call besides waiting for its
workers.  With N workers on one GPU the workers' own time does not shrink (they share the card); the parent's share is what must stay
small for N real GPUs to scale.  Prints one JSON line per N.   usage: multi_run.py [--reads 10000000] [--workers 1,2,4,6] [--check]"""
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
    ap.add_argument("--reads", type=int, default=10000000)
    ap.add_argument("--workers", default="1,2,4,6")
    ap.add_argument("--check", action="store_true", help="coordinates of every N == those of N = 1")
    args = ap.parse_args()
    import synth
    from bench import its2_profiles
    from itsxpress_amd.multi import MultiEngine
    from itsxpress_amd.trim import write_trimmed_fastq
    with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f:
        thmm = f.read()
    hmm = its2_profiles(thmm)
    n = args.reads
    blob, offs = synth.make_reads(thmm, n, config=3, seed=synth.SEED + 3, fixed_len=0, len_range=(300, 580))
    tmp = tempfile.mkdtemp(prefix="itsx_multi_run_")
    ref = None
    try:
        plain = os.path.join(tmp, "in.fastq")
        bases = np.frombuffer(blob, np.uint8)
        with open(plain, "wb") as f:
            q = b"I" * 600
            for i in range(n):
                s = bases[offs[i]:offs[i + 1]]
                f.write(b"@read%d 1:N:0:1\n" % i + s.tobytes() + b"\n+\n" + q[:len(s)] + b"\n")
        fq = os.path.join(tmp, "in.fastq.gz")
        write_trimmed_fastq(plain, fq, np.zeros(n, np.int32), np.full(n, 1 << 30, np.int32), gzipped=True)
        os.remove(plain)
        del blob, bases
        for N in [int(x) for x in args.workers.split(",")]:
            from itsxpress_amd import _lib
            _lib.lib().itsx_io_cache_clear()               # every N inflates the file itself
            t0 = time.perf_counter()
            me = MultiEngine(N, devices=[0] * N)
            t_start = time.perf_counter() - t0
            wall = {}
            try:
                me.set_rows_mode("lazy")
                for name, fn in (("load_reads_file", lambda: me.load_reads_file(fq)), ("derep", lambda: me.derep()),
                                 ("load_profiles", lambda: me.load_profiles(text=hmm)), ("search", lambda: me.search()),
                                 ("finalize", lambda: me.finalize()), ("trim_coords", lambda: me.trim_coords("3_", "4_"))):
                    t0 = time.perf_counter()
                    out = fn()
                    wall[name] = time.perf_counter() - t0
                coords = out
                st = me.stats()
                # the workers' own device time for the step (each one's share; on N real GPUs they run side by side)
                dev_ms = [s["ms_derep"] + s["ms_msv"] + s["ms_filters"] + s["ms_domains"] + s["ms_finalize"] for s in st]
                line = {"workers": N, "reads": n, "unique": int(me.n_unique), "wall_s": {k: round(v, 3) for k, v in wall.items()},
                        "wall_total_s": round(sum(wall.values()), 3), "workers_started_s": round(t_start, 2),
                        "parent_s": {k: round(v, 3) for k, v in me.parent_s.items()},
                        "parent_own_s": {k: round(v, 3) for k, v in me.parent_own_s.items()},
                        "parent_own_compute_path_s": round(sum(v for k, v in me.parent_own_s.items() if not k.startswith("load")), 3),
                        "worker_device_ms": [round(x) for x in dev_ms],
                        # round 6: the parent deals byte ranges of a shared text while it is still inflating the rest -- seconds from the call to the
                        # first worker having its range, and the parent's serial share of the wall (its own work outside the load + that lead time)
                        "first_worker_busy_s": round(getattr(me, "first_worker_busy_s", float("nan")) or 0.0, 3),
                        "parent_serial_share_of_wall": round((sum(v for k, v in me.parent_own_s.items() if not k.startswith("load")) + (getattr(me, "first_worker_busy_s", 0.0) or 0.0)) / max(1e-9, sum(wall.values())), 4),
                        "note": "parent_s: wall time of this process inside a call; parent_own_s: the part that was its own work, not a wait for the workers "
                                "('load: inflate + cut' = inflate the file once, cut it, write the pieces; derep and trim_coords are three commands each); "
                                "parent_own_compute_path_s = everything but the load; N workers share ONE GPU here"}
                if args.check:
                    if ref is None:
                        ref = coords
                    line["coordinates_equal_one_worker"] = bool(all(np.array_equal(a, b) for a, b in zip(coords, ref)))
                    assert line["coordinates_equal_one_worker"]
                print(json.dumps(line), flush=True)
            finally:
                me.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
