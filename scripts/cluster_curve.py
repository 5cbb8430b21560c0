#!/usr/bin/env python3
"""reads -> seconds curve of row a2 (greedy clustering, id 0.995) on BASELINE configs[4]-shaped reads: 2x250-merged lengths
300-480, `1_` / `4_` motifs (--region ALL), labels in input order.  usage: cluster_curve.py --sizes 200000,1000000 [--id 0.995]
Prints one JSON line per size; with --search the HMM stages run on the centroids too (1_ / 4_ profiles)."""
import argparse
import gzip
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def region_all_profiles(hmm_text):
    blocks = [b + "//\n" for b in hmm_text.split("//\n") if "NAME  " in b]
    return "".join(b for b in blocks if b.split("NAME  ")[1][:2] in ("1_", "4_"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="200000,1000000")
    ap.add_argument("--id", type=float, default=0.995)
    ap.add_argument("--search", action="store_true")
    ap.add_argument("--passes", type=int, default=1)
    args = ap.parse_args()
    import synth
    from itsxpress_amd import Engine
    with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f:
        thmm = f.read()
    eng = Engine(0)
    if args.search:
        eng.load_profiles(text=region_all_profiles(thmm))
    for n in [int(x) for x in args.sizes.split(",")]:
        blob, offs = synth.make_reads(thmm, n, config=4, left="1_", right="4_", fixed_len=0, len_range=(300, 480), as_array=True)
        for p in range(args.passes):
            eng.set_reads_buffer(blob, offs)
            t0 = time.perf_counter()
            eng.cluster(args.id, strand_both=True)
            t_cl = time.perf_counter() - t0
            t_se = None
            if args.search:
                t0 = time.perf_counter()
                eng.search(T=10.0, F1=1e-6, F2=1e-6, F3=1e-6)
                eng.finalize(domE=10.0)
                eng.trim_coords("1_", "4_")
                t_se = time.perf_counter() - t0
            st = eng.stats()
            print(json.dumps({"reads": n, "pass": p, "id": args.id, "cluster_s": round(t_cl, 3), "search_s": t_se, "centroids": int(st["n_unique"]),
                              "windows": int(st["cl_windows"]), "cut_windows": int(st["cl_cuts"]), "full_alignments": int(st["cl_alignments"]),
                              "certified": int(st["cl_certified"]), "mean_len": float(offs[-1]) / n}), flush=True)


if __name__ == "__main__":
    main()
