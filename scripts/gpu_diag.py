#!/usr/bin/env python3
"""Stage-by-stage diagnostic on a GPU box: engine vs oracle, printing mismatch summaries
instead of stopping at the first assert.  Usage: python scripts/gpu_diag.py [n_synth_reads]"""
import gzip
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
import synth  # noqa: E402
from itsxpress_amd import Engine  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def cmp_field(name, a, b, isfloat=False, mask=None):
    if mask is not None:
        a, b = a[mask], b[mask]
    if isfloat:
        bad = bits(a) != bits(b)
    else:
        bad = a != b
    n = int(bad.sum())
    msg = "  %-14s mismatches %d / %d" % (name, n, len(a))
    if n:
        idx = np.nonzero(bad)[0][:5]
        msg += "   first: " + ", ".join("%d:(%r vs %r)" % (i, a[i].item(), b[i].item()) for i in idx)
        if isfloat:
            msg += "  max|diff| %.3g" % float(np.nanmax(np.abs(a[bad].astype(np.float64) - b[bad].astype(np.float64))))
    print(msg)
    return n


def run(eng, hmm, seqs, label, left="3_", right="4_"):
    print("==== %s: %d reads" % (label, len(seqs)))
    t0 = time.time()
    eng.load_profiles(text=hmm)
    eng.set_reads(seqs)
    nu = eng.derep()
    rep_of, strand, uniq_of = eng.get_derep()
    codes, o = orc.digitize(seqs)
    nc, orep, ostrand = orc.derep(codes, o)
    print("derep: engine %d oracle %d unique; rep_of mismatches %d strand mismatches %d" %
          (nu, nc, int((rep_of != orep).sum()), int((strand != ostrand).sum())))
    eng.search()
    eng.finalize()
    t1 = time.time()
    st = eng.stats()
    print("engine wall %.2fs  stats %s" % (t1 - t0, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}))
    seed, _ = eng.get_uniques()
    useqs = [seqs[int(i)] for i in seed]
    c2, o2 = orc.digitize(useqs)
    t2 = time.time()
    res = orc.SearchResult(orc.HmmSet(text=hmm), c2, o2, threads=os.cpu_count() or 8, keep_trace=1)
    print("oracle wall %.2fs counts %s ndom %d" % (time.time() - t2, res.counts, len(res.domains)))
    tr, ot = eng.pairtraces(), res.trace
    print("traces: engine %d oracle %d" % (len(tr), len(ot)))
    total = 0
    if len(tr) == len(ot) and np.array_equal(tr["rep"], ot["seq"]) and np.array_equal(tr["prof"], ot["prof"]):
        total += cmp_field("msv_xj", tr["msv_xj"], ot["msv_xj"])
        for f in ("msv_sc", "nullsc", "filtersc"):
            total += cmp_field(f, tr[f], ot[f], True)
        total += cmp_field("pass_bias", tr["pass_bias"], ot["pass_bias"])
        pb = (ot["pass_bias"] == 1) & (tr["pass_bias"] == 1)
        total += cmp_field("fwdsc", tr["fwdsc"], ot["fwdsc"], True, pb)
        total += cmp_field("pass_fwd", tr["pass_fwd"], ot["pass_fwd"])
        pf = (ot["pass_fwd"] == 1) & (tr["pass_fwd"] == 1)
        total += cmp_field("bcksc", tr["bcksc"], ot["bcksc"], True, pf)
        total += cmp_field("nregions", tr["nregions"], ot["nregions"], False, pf)
        total += cmp_field("ndom", tr["ndom"], ot["ndom"], False, pf)
    else:
        total += 1
        es = set(zip(tr["prof"].tolist(), tr["rep"].tolist()))
        os_ = set(zip(ot["prof"].tolist(), ot["seq"].tolist()))
        print("  MSV survivor sets differ: only-engine %d only-oracle %d" % (len(es - os_), len(os_ - es)))
        print("   e.g. only engine", list(es - os_)[:5], "only oracle", list(os_ - es)[:5])
    d, od = eng.domains(), res.domains
    print("domains: engine %d oracle %d" % (len(d), len(od)))
    if len(d) == len(od):
        for f in ("rep", "prof", "tlen", "ienv", "jenv", "dom_idx", "ndom", "seq_reported", "dom_reported"):
            total += cmp_field(f, d[f], od["seq" if f == "rep" else f])
        for f in ("envsc", "domcorrection", "dombias", "bitscore", "seq_score", "seq_bias"):
            total += cmp_field(f, d[f], od[f], True)
    else:
        total += 1
    got, exp = eng.rep_coords(left, right), res.positions(left, right)
    for nm, g, e in zip(("start", "stop", "tlen", "in_ddict"), got, exp):
        total += cmp_field(nm, g, e)
    print("==== %s: TOTAL MISMATCHES %d" % (label, total))
    return total


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    eng = Engine(0)
    names, seqs = [], []
    with gzip.open(os.path.join(GOLD, "fixture_reads.fa.gz"), "rt") as f:
        for line in f:
            (names if line[0] == ">" else seqs).append(line[1:].strip() if line[0] == ">" else line.strip())
    mini = open(os.path.join(GOLD, "mini.hmm")).read()
    tot = run(eng, mini, seqs, "fixture x mini.hmm")
    thmm = gzip.open(os.path.join(GOLD, "T.hmm.gz"), "rt").read()
    blocks = [b + "//\n" for b in thmm.split("//\n") if "NAME  " in b]
    its2 = "".join(b for b in blocks if b.split("NAME  ")[1][:2] in ("3_", "4_"))
    blob, offs = synth.make_reads(thmm, n, seed=21)
    tot += run(eng, its2, synth.to_strings(blob, offs), "synthetic x T ITS2")
    print("ALL DONE total mismatches", tot)
    return 0 if tot == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
