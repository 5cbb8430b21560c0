#!/usr/bin/env python3
"""What would prefix-sharing Forward save on the bench workload?  (host-only, ~5 min, ~15 GB.)  Generates BASELINE
configs[2]'s 10 M reads, takes the distinct ones in (length, lexicographic) order, cuts them into 64-lane waves and forms
leader / follower groups greedily (a group's followers start Forward at the group's common prefix from the leader's row state);
prints the share of Forward rows the followers would skip.  Result at 10 M reads: 4.8 % (4.4 % when a random 74 % of the uniques
pass MSV for a profile) -- too little to pay for the sort, the grouping and a second kernel launch per batch."""
import sys, time, gzip, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10_000_000
W = 448
thmm = gzip.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'T.hmm.gz'), 'rt').read()
t0 = time.time()
blob, offs = synth.make_reads(thmm, N, config=3, fixed_len=0, len_range=(300, 580), as_array=True)
lens = np.diff(offs)
print("generated", time.time() - t0, flush=True)
# padded matrix [N, 2 + W]: big-endian length, then the first W bases
M = np.zeros((N, 2 + W), np.uint8)
M[:, 0] = lens >> 8; M[:, 1] = lens & 255
col = np.arange(W)
BLK = 1 << 18
for c0 in range(0, N, BLK):
    c1 = min(N, c0 + BLK)
    ln = np.minimum(lens[c0:c1], W)
    idx = offs[c0:c1, None] + col[None, :]
    mask = col[None, :] < ln[:, None]
    sub = M[c0:c1, 2:]
    sub[mask] = blob[idx[mask]]
print("padded", time.time() - t0, flush=True)
S = M.view('S%d' % (2 + W)).ravel()
U = np.unique(S)
print("uniques", len(U), time.time() - t0, flush=True)
A = U.view(np.uint8).reshape(len(U), 2 + W)
Lu = (A[:, 0].astype(np.int64) << 8) | A[:, 1]
# adjacent lcp (bases), 0 across different lengths
lcp = np.zeros(len(U), np.int64)
for c0 in range(1, len(U), BLK):
    c1 = min(len(U), c0 + BLK)
    d = A[c0:c1, 2:] != A[c0 - 1:c1 - 1, 2:]
    first = np.where(d.any(axis=1), d.argmax(axis=1), W)
    same_len = Lu[c0:c1] == Lu[c0 - 1:c1 - 1]
    lcp[c0:c1] = np.where(same_len, np.minimum(first, np.minimum(Lu[c0:c1], W)), 0)
print("lcp done", time.time() - t0, flush=True)
# ---- the TREE figure (round 5): every distinct prefix of a length group is computed once.  Rows shared = sum of the common prefix with
# the lexicographic predecessor of equal length (the trie's internal path lengths), exact and with row-state check-points every B rows;
# branch nodes = distinct (depth, prefix) check-points some later read starts from (each needs one saved row state per profile).
tot = int(Lu.sum())
print(json.dumps({"uniques": int(len(U)), "sum_L": tot, "rows_shared_frac_exact": round(float(lcp.sum()) / tot, 4)}), flush=True)
for B in (16, 32, 64, 128):
    lb = lcp // B
    shared = int((lb * B).sum())
    nodes = 0; chains_by_depth = []
    for d in range(1, int(lb.max()) + 1):
        ids = np.cumsum(lb < d)              # node number at depth d of every read's prefix
        sel = lb == d
        chains_by_depth.append(int(sel.sum()))
        nodes += len(np.unique(ids[sel]))
    print(json.dumps({"B": B, "rows_shared_frac": round(shared / tot, 4), "branch_nodes": nodes, "roots": int((lb == 0).sum()),
                      "chains_by_start_depth": chains_by_depth[:24]}), flush=True)
if "--tree-only" in sys.argv:
    sys.exit(0)
for frac in (1.0, 0.74):
    rng = np.random.default_rng(1)
    keep = np.flatnonzero(rng.random(len(U)) < frac) if frac < 1 else np.arange(len(U))
    # lcp between kept neighbours = min of adjacent lcps in between
    # compute via running minimum over gaps
    run = np.minimum.reduceat(np.concatenate([lcp, [0]]), np.concatenate([[0], keep[:-1] + 1])) if False else None
    l2 = np.zeros(len(keep), np.int64)
    if frac < 1:
        # min over lcp[keep[i-1]+1 .. keep[i]]
        starts = keep[:-1] + 1
        ends = keep[1:] + 1
        cm = np.minimum.reduceat(lcp, np.stack([starts, ends], 1).ravel()[:-1])[::2] if len(keep) > 1 else np.array([], np.int64)
        l2[1:] = cm
    else:
        l2 = lcp.copy()
    Lk = Lu[keep]
    nw = len(keep) // 64
    l2w = l2[:nw * 64].reshape(nw, 64)
    p_w = l2w[:, 1:].min(axis=1)          # common prefix inside the wave
    c_w = l2w[:, 0]                        # to the previous wave's last lane
    Lw = Lk[:nw * 64].reshape(nw, 64)[:, 0]
    # greedy grouping
    saved = 0; total = int(Lw.sum()); w = 0; groups = 0
    pw = p_w.tolist(); cw = c_w.tolist()
    while w < nw:
        r = pw[w]; best = 0; bestT = 1; T = 1; v = w + 1
        while v < nw and T < 4096:
            r = min(r, cw[v], pw[v])
            if r < 32: break
            T += 1
            ben = (T - 1) * r
            if ben > best: best = ben; bestT = T
            v += 1
        saved += best; groups += bestT > 1
        w += bestT
    print("pass fraction %.2f: waves %d, groups %d, Forward rows saved %.1f %%" % (frac, nw, groups, 100.0 * saved / total), flush=True)
