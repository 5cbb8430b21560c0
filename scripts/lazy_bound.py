#!/usr/bin/env python3
"""Experiment behind the lazy domain stage (round 4): how many (representative, profile) pairs past the Forward filter can
never win ItsPosition's argmax?  Runs on the CPU ORACLE (test infrastructure) on a sample of the bench workload.

The bound.  For a pair past Forward we know fwdsc (multihit Forward of the whole target, nats), nullsc and the target length
n.  hmmsearch's per-domain bit score (p7_pipeline.c, restated in oracle/orc_search.c) is

    bits = (envsc + (n - Ld) ln(n / (n + 3)) - (nullsc + dombias)) / ln 2,    dombias = logsum(0, ln omega + domcorr) >= 0,

with envsc the UNIHIT Forward score of the envelope x_i..x_j (Ld residues) under the length model of the whole target
(loop n / (n + 2), move 2 / (n + 2), E->C = 1).  Every path of that unihit sum -- a residues in N, the core, b residues in
C, a + b <= Ld -- is also a path of the multihit Forward of the whole target with the flanks in N and C: odds
loop_m^(n - Ld + a + b) move_m^2 (1/2) core, loop_m = n / (n + 3), move_m = 3 / (n + 3).  Path by path

    unihit / multihit  =  (loop_u / loop_m)^(a + b) (move_u / move_m)^2 2 / loop_m^(n - Ld)
                       <= ((n + 3) / (n + 2))^Ld (2 (n + 3) / (3 (n + 2)))^2 2 / loop_m^(n - Ld),

hence  envsc + (n - Ld) ln loop_m  <=  fwdsc + ln 2 + 2 ln(2 (n + 3) / (3 (n + 2))) + Ld ln((n + 3) / (n + 2)),  and with
Ld <= n and dombias >= 0

    bits <= (fwdsc - nullsc) / ln 2 + C(n),   C(n) = 1 + (2 ln(2 (n + 3) / (3 (n + 2))) + n ln((n + 3) / (n + 2))) / ln 2.

(A clustered region's envelopes are unihit Forward scores of sub-intervals too, with another null2: the same bound.)
Float rounding: both sides are sums of positive products of <= ~3 (n + M) rounded operations per path; relative error
<= 3 (n + M) 2^-24 ~ 1e-4 -> 2e-4 bits at n = 600; MARGIN = 0.02 bits covers it a hundred times.

The experiment: per (representative, class = 2-character profile prefix) the winner = best reported domain's %.1f score in
tenths; a pair is prunable iff round-up(10 (bound + MARGIN)) < winner tenths (ties go to the EARLIER row, so an equal bound
must be evaluated).  Prints the pruning rate and the histogram of bound - winner; `--json` appends a record.
"""
import argparse
import gzip
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MARGIN = 0.02
LN2 = math.log(2.0)


def c_of_n(n):
    n = np.asarray(n, np.float64)
    return 1.0 + (2.0 * np.log(2.0 * (n + 3.0) / (3.0 * (n + 2.0))) + n * np.log((n + 3.0) / (n + 2.0))) / LN2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--workload", choices=["cfg2", "cfg1", "cfg3"], default="cfg2")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 4)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import orc
    import synth
    with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f:
        thmm = f.read()
    blocks = [b + "//\n" for b in thmm.split("//\n") if "NAME  " in b]
    hmm = "".join(b for b in blocks if b.split("NAME  ")[1][:2] in ("3_", "4_"))
    if args.workload == "cfg3":
        with gzip.open(os.path.join(ROOT, "tests", "golden", "all_its2.hmm.gz"), "rt") as f:
            hmm = f.read()
    gen = dict(config=3, seed=synth.SEED + 3, left="3_", right="4_")
    if args.workload != "cfg1":
        gen.update(fixed_len=0, len_range=(300, 580))
    blob, offs = synth.make_reads(thmm, args.reads, **gen)
    seqs = synth.to_strings(blob, offs)
    orc.use_library("libbase_sse.so")
    hs = orc.HmmSet(text=hmm)
    codes, o = orc.digitize(seqs)
    nc, rep, strand = orc.derep(codes, o)
    seeds = [i for i in range(len(seqs)) if rep[i] == i]
    c2, o2 = orc.digitize([seqs[i] for i in seeds])
    t0 = time.time()
    res = orc.SearchResult(hs, c2, o2, threads=args.threads, keep_trace=1)
    dt = time.time() - t0
    tr, dom = res.trace, res.domains
    P = hs.n
    cls_names = sorted(set(n[:2] for n in hs.names))
    cls_of_prof = np.array([cls_names.index(n[:2]) for n in hs.names])
    ulen = np.diff(o2)
    pf = tr[tr["pass_fwd"] != 0]
    n_pf = len(pf)
    L = ulen[pf["seq"]]
    bound = (pf["fwdsc"].astype(np.float64) - pf["nullsc"].astype(np.float64)) / LN2 + c_of_n(L)
    key_pf = pf["seq"] * P + pf["prof"]
    order = np.argsort(key_pf)
    key_sorted = key_pf[order]
    # (1) the bound holds for every domain row
    dk = dom["seq"] * P + dom["prof"]
    pos = np.searchsorted(key_sorted, dk)
    assert (key_sorted[pos] == dk).all()
    b_of_dom = bound[order][pos]
    slack = b_of_dom - dom["bitscore"].astype(np.float64)
    viol = int((slack < 0).sum())
    # (2) winners per (representative, class): best REPORTED domain, %.1f tenths
    tenths = np.rint(dom["bitscore"].astype(np.float64) * 10.0).astype(np.int64)
    rep_ok = dom["dom_reported"] != 0
    gkey_dom = dom["seq"] * len(cls_names) + cls_of_prof[dom["prof"]]
    nG = len(seeds) * len(cls_names)
    win = np.full(nG, -10 ** 9, np.int64)
    np.maximum.at(win, gkey_dom[rep_ok], tenths[rep_ok])
    # certain winners only (what k_compact_* calls certain: exp(lnP) * 1e9 <= 0.01)
    certain = rep_ok & (dom["lnP"] <= math.log(0.01 / 1e9) - 1e-6)
    winc = np.full(nG, -10 ** 9, np.int64)
    np.maximum.at(winc, gkey_dom[certain], tenths[certain])
    gkey_pf = pf["seq"] * len(cls_names) + cls_of_prof[pf["prof"]]
    b10 = np.ceil((bound + MARGIN) * 10.0 - 0.5).astype(np.int64)       # largest tenths value a domain of the pair can print
    prunable = b10 < winc[gkey_pf]
    prunable_any = b10 < win[gkey_pf]
    has_w = winc[gkey_pf] > -10 ** 9
    diff = (bound - win[gkey_pf] / 10.0)[win[gkey_pf] > -10 ** 9]
    hist, edges = np.histogram(diff, bins=[-1e9, -40, -30, -20, -10, -5, -2, -1, 0, 1, 2, 5, 1e9])
    # the pair that holds the winner must be evaluated; rounds: evaluate the top-bound pair per group first
    # how many pairs does a 2-round schedule evaluate?  round 1 = the best-bound pair of each group; round 2 = everything with b10 >= winner found
    # (approximated here by the final certain winner: the schedule converges to it)
    n_groups = int((np.bincount(gkey_pf, minlength=nG) > 0).sum())
    evaluated = int((~prunable).sum())
    rec = {
        "workload": args.workload, "reads": args.reads, "unique": len(seeds), "profiles": P, "oracle_s": round(dt, 1),
        "pairs_past_fwd": n_pf, "domain_rows": int(len(dom)), "bound_violations": viol,
        "min_slack_bits": float(slack.min()) if len(slack) else None, "median_slack_bits": float(np.median(slack)) if len(slack) else None,
        "groups_with_pairs": n_groups, "groups_with_certain_winner": int((winc > -10 ** 9).sum()),
        "pairs_in_groups_with_certain_winner": int(has_w.sum()),
        "pairs_prunable_vs_certain_winner": int(prunable.sum()), "prune_rate": float(prunable.mean()) if n_pf else None,
        "pairs_prunable_vs_any_winner": int(prunable_any.sum()),
        "pairs_to_evaluate": evaluated, "evaluate_rate": evaluated / max(n_pf, 1),
        "hist_bound_minus_winner_bits": {"edges": [float(e) for e in edges[1:-1]], "counts": [int(x) for x in hist]},
        "C_of_n": {"300": float(c_of_n(300)), "440": float(c_of_n(440)), "580": float(c_of_n(580))}, "margin_bits": MARGIN,
    }
    print(json.dumps(rec, indent=1))
    if args.json:
        with open(args.json, "a") as f:
            f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
