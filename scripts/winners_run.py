#!/usr/bin/env python3
"""ITSXPRESS_DOMTBL=winners at scale: N synthetic merged reads (configs[2]'s shape) -> derep -> lazy search -> uc.txt, rep.fa and the
kept-rows domtbl.txt as FILES (what the reference's own Dedup / ItsPosition parse), stage by stage; --full-reads M runs the default
(full-table) mode on the first M reads beside it for the row / byte / time ratios; --parse times the mirror's ItsPosition on the file.
Prints one JSON line.  usage: winners_run.py [--reads 10000000] [--full-reads 1000000] [--parse]"""
import argparse
import gzip
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(eng, blob, offs, mode, tmp, tag, parse):
    from itsxpress_amd import ItsPosition
    t = {}
    t0 = time.perf_counter()
    eng.set_rows_mode("lazy" if mode == "winners" else "full")
    eng.set_reads_buffer(blob, offs)
    eng.derep(strand_both=True, minseqlength=1)
    t["load+derep"] = time.perf_counter() - t0
    t1 = time.perf_counter()
    eng.search(T=10.0, F1=1e-6, F2=1e-6, F3=1e-6)
    eng.finalize(domE=10.0)
    t["search+finalize"] = time.perf_counter() - t1
    t1 = time.perf_counter()
    uc, rep, dom = (os.path.join(tmp, "%s_%s" % (tag, n)) for n in ("uc.txt", "rep.fa", "domtbl.txt"))
    eng.write_uc(uc)
    eng.write_rep_fasta(rep)
    t["write uc.txt + rep.fa"] = time.perf_counter() - t1
    t1 = time.perf_counter()
    eng.set_kept_rows(mode == "winners")
    eng.write_domtbl(dom)
    t["write domtbl.txt"] = time.perf_counter() - t1
    out = {"mode": mode, "reads": len(offs) - 1, "unique": eng.n_unique, "stages_s": {k: round(v, 3) for k, v in t.items()},
           "s_total": round(time.perf_counter() - t0, 3), "domtbl_MB": round(os.path.getsize(dom) / 1e6, 1),
           "uc_MB": round(os.path.getsize(uc) / 1e6, 1), "rep_MB": round(os.path.getsize(rep) / 1e6, 1)}
    rows = 0
    with open(dom, "rb") as f:
        for ln in f:
            rows += ln[:1] != b"#"
    out["domtbl_rows"] = rows
    if parse:
        t1 = time.perf_counter()
        d = ItsPosition(dom, "ITS2").ddict
        out["ItsPosition_parse_s"] = round(time.perf_counter() - t1, 1)
        out["targets_with_a_row"] = len(d)
    for p in (uc, rep, dom):
        os.remove(p)
    eng.set_kept_rows(False)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10000000)
    ap.add_argument("--full-reads", type=int, default=1000000)
    ap.add_argument("--parse", action="store_true")
    ap.add_argument("--dir", default="", help="where the files go (default: a temporary directory under /dev/shm when it exists)")
    args = ap.parse_args()
    import synth
    from bench import its2_profiles
    from itsxpress_amd import Engine
    with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f:
        thmm = f.read()
    blob, offs = synth.make_reads(thmm, args.reads, config=3, seed=synth.SEED + 3, fixed_len=0, len_range=(300, 580), as_array=True)
    tmp = tempfile.mkdtemp(prefix="itsx_winners_", dir=args.dir or ("/dev/shm" if os.path.isdir("/dev/shm") else None))
    eng = Engine(0)
    res = {}
    try:
        eng.load_profiles(text=its2_profiles(thmm))
        run(eng, blob[:int(offs[20000])], offs[:20001], "winners", tmp, "warm", False)        # first-use allocations
        res["winners"] = run(eng, blob, offs, "winners", tmp, "w", args.parse)
        print("[winners_run] winners: %s" % json.dumps(res["winners"]), file=sys.stderr, flush=True)
        if args.full_reads > 0:
            m = min(args.full_reads, args.reads)
            res["winners_on_the_full_leg's_reads"] = run(eng, blob[:int(offs[m])], offs[:m + 1], "winners", tmp, "ws", args.parse)
            res["full"] = run(eng, blob[:int(offs[m])], offs[:m + 1], "full", tmp, "f", args.parse)
    finally:
        eng.close()
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
