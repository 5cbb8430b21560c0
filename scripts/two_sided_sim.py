#!/usr/bin/env python3
"""What does TWO-SIDED sharing leave of pass A's rows on the bench workload?  (host-only, numpy.)

Models the device schedule of csrc/k_share.hip as built in round 6, per length group of the distinct reads (in the engine's
order: by length, then by first occurrence), over blocks of B rows:
  * PREFIX tree: node (d, first d*B residues); owner = first unique with them.  fd[s] = deepest node s does not own (its Forward
    chain starts there from the owner's saved state);
  * SUFFIX tree: node (r, last r blocks: rows (A-r)*B+1 .. L, A = ceil(L/B)); owner likewise.  rd[s] = deepest suffix node s does
    not own;
  * JOIN: a unique with rd > 0 stops its Forward chain at level j = max(fd, A - rd) and takes the rest of the sum over paths
    from the Backward state the owner of its suffix node at level j saved there (score = <alpha_j, gamma_j>);
  * a Forward chain runs on to the deepest level a prefix-child starts from (it must save the state there); a unique without
    a shared suffix runs to L as before;
  * a Backward chain exists only where somebody joins (or a suffix-child that runs starts): from its own start (A - rd, or L
    for a suffix root) down to the lowest level it must save at.
Prints rows computed / sum of L for today's tree (prefix only) and for the two-sided schedule, the saved-state counts and the
distribution of chain lengths.  The review's figure to reproduce (2 M reads, B = 32): 0.59 -> 0.28.

usage: two_sided_sim.py [N_reads] [--B 16,32,64]"""
import sys, time, gzip, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2_000_000
Bs = (16, 32, 64)
if "--B" in sys.argv:
    Bs = tuple(int(x) for x in sys.argv[sys.argv.index("--B") + 1].split(","))
thmm = gzip.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'T.hmm.gz'), 'rt').read()
t0 = time.time()
blob, offs = synth.make_reads(thmm, N, config=3, fixed_len=0, len_range=(300, 580), as_array=True)
lens = np.diff(offs).astype(np.int64)
print("generated %d reads in %.0f s" % (N, time.time() - t0), flush=True)

rng = np.random.default_rng(7)
MULT = rng.integers(1, 2 ** 63, size=1024, dtype=np.uint64) | np.uint64(1)


def first_owner(keys):
    """index of the first element with the same key, for every element"""
    order = np.argsort(keys, kind="stable")
    ks = keys[order]
    newgrp = np.ones(len(ks), bool); newgrp[1:] = ks[1:] != ks[:-1]
    first = order[np.flatnonzero(newgrp)]                       # stable sort: the first of a run is the smallest index
    gid = np.cumsum(newgrp) - 1
    own = np.empty(len(keys), np.int64)
    own[order] = first[gid]
    return own


def group_stats(X, L, B):
    """X: [n, L] uint8 distinct rows in first-occurrence order.  Returns a dict of row / node counts."""
    n = X.shape[0]
    A = (L + B - 1) // B
    hasN = (X == ord('N'))
    idx = np.arange(L)
    firstN = np.where(hasN.any(1), hasN.argmax(1), 1 << 30)
    lastN = np.where(hasN.any(1), L - 1 - hasN[:, ::-1].argmax(1), -1)
    # block hashes (polynomial over the block's bytes)
    pad = A * B - L
    Xp = np.concatenate([X, np.zeros((n, pad), np.uint8)], 1).reshape(n, A, B).astype(np.uint64)
    bh = (Xp * MULT[None, None, :B]).sum(2, dtype=np.uint64)    # [n, A]
    me = np.arange(n)
    # ---- prefix tree
    dlim = np.minimum(np.minimum((L - 1) // B, 63), firstN // B)
    fd = np.zeros(n, np.int64); fpar = np.full(n, -1, np.int64)
    h = np.full(n, 0x9E3779B97F4A7C15, np.uint64)
    for d in range(1, A):
        h = (h * np.uint64(0x100000001B3) + bh[:, d - 1]) * np.uint64(0xD6E8FEB86659FD93)
        ok = dlim >= d
        own = me.copy()
        if ok.sum() > 1:
            sub = np.flatnonzero(ok)
            own[sub] = sub[first_owner(h[sub])]
        sel = own != me
        fd[sel] = d; fpar[sel] = own[sel]
    # ---- suffix tree (r = blocks from the end; block A-1 is the partial one)
    rlim = np.minimum(np.minimum(A - 1, 63), A - 1 - np.maximum(lastN, -1) // B - (lastN >= 0) * 0)
    rlim = np.where(lastN >= 0, np.minimum(np.minimum(A - 1, 63), A - 1 - lastN // B), np.minimum(A - 1, 63))
    rd = np.zeros(n, np.int64)
    rown = np.full((n, A), -1, np.int64)                       # owner of the suffix node at every r (for the join level)
    h = np.full(n, 0xC2B2AE3D27D4EB4F, np.uint64)
    for r in range(1, A):
        h = (h * np.uint64(0x100000001B3) + bh[:, A - r]) * np.uint64(0xD6E8FEB86659FD93)
        ok = rlim >= r
        own = me.copy()
        if ok.sum() > 1:
            sub = np.flatnonzero(ok)
            own[sub] = sub[first_owner(h[sub])]
        rown[:, r] = np.where(ok, own, -1)
        sel = own != me
        rd[sel] = r
    # ---- joins
    sd = A - rd                                                   # suffix level of the deepest shared node (A: none)
    joins = rd > 0
    j = np.where(joins, np.maximum(fd, sd), A)                   # join level (A = runs to L)
    rj = A - j                                                    # in suffix coordinates
    jown = np.where(joins, rown[me, np.maximum(rj, 0)], -1)
    bad = joins & ((jown < 0) | (jown == me))
    joins &= ~bad; j[bad] = A
    # forward save masks / chain ends
    maxchild = np.zeros(n, np.int64)
    np.maximum.at(maxchild, fpar[fd > 0], fd[fd > 0])
    fend_lvl = np.where(joins, np.maximum(j, maxchild), A)
    fend_row = np.where(joins, np.minimum(fend_lvl * B, L), L)
    frows = fend_row - fd * B
    fnodes = len(np.unique(fpar[fd > 0] * 64 + fd[fd > 0])) if (fd > 0).any() else 0
    # backward: save masks by propagation (deepest r first)
    rmask = np.zeros(n, np.uint64)
    if joins.any():
        np.bitwise_or.at(rmask, jown[joins], (np.uint64(1) << rj[joins].astype(np.uint64)))
    rpar = np.where(rd > 0, rown[me, rd], -1)
    for r in range(int(rd.max()), 0, -1):
        sel = (rd == r) & (rmask != 0)
        if sel.any():
            np.bitwise_or.at(rmask, rpar[sel], np.uint64(1) << np.uint64(r))
    runs = rmask != 0
    rend = np.zeros(n, np.int64)
    m = rmask.copy()
    for b in range(63, 0, -1):
        hit = (rend == 0) & ((m >> np.uint64(b)) & np.uint64(1)).astype(bool)
        rend[hit] = b
    # rows of a backward chain: from level A - rd (row min(L, (A - rd) * B)) down to level A - rend
    bstart_row = np.minimum((A - rd) * B, L)
    brows = np.where(runs, bstart_row - (A - rend) * B, 0)
    assert (brows >= 0).all()
    bnodes = int(sum(bin(int(x)).count("1") for x in rmask[runs]))
    own_blocks = (frows + B - 1) // B
    return dict(sumL=n * L, rows_prefix=int((L - fd * B).sum()), frows=int(frows.sum()), brows=int(brows.sum()), fnodes=fnodes, bnodes=bnodes,
                joins=int(joins.sum()), n=n, bchains=int(runs.sum()), zero_row=int((frows == 0).sum()),
                hist=np.bincount(np.minimum(own_blocks, 24), minlength=25))


# distinct reads in first-occurrence order, grouped by length
res = {B: None for B in Bs}
order = np.argsort(lens, kind="stable")
ls = lens[order]
cuts = np.flatnonzero(np.diff(ls)) + 1
starts = np.concatenate([[0], cuts]); ends = np.concatenate([cuts, [len(ls)]])
tot = {B: dict(sumL=0, rows_prefix=0, frows=0, brows=0, fnodes=0, bnodes=0, joins=0, n=0, bchains=0, zero_row=0, hist=np.zeros(25, np.int64)) for B in Bs}
col = None
for gi, (a, b) in enumerate(zip(starts, ends)):
    L = int(ls[a]); ids = order[a:b]
    X = blob[offs[ids, None] + np.arange(L)[None, :]]
    S = np.ascontiguousarray(X).view('S%d' % L).ravel()
    _, first = np.unique(S, return_index=True)
    first.sort()
    X = X[first]
    for B in Bs:
        st = group_stats(X, L, B)
        for k, v in st.items():
            tot[B][k] = tot[B][k] + v
    if gi % 40 == 0:
        print("length %d: %d distinct of %d (%.0f s)" % (L, len(first), b - a, time.time() - t0), flush=True)
for B in Bs:
    t = tot[B]
    out = {"reads": N, "uniques": int(t["n"]), "B": B,
           "rows_prefix_only_frac": round(t["rows_prefix"] / t["sumL"], 4),
           "rows_two_sided_frac": round((t["frows"] + t["brows"]) / t["sumL"], 4),
           "forward_frac": round(t["frows"] / t["sumL"], 4), "backward_frac": round(t["brows"] / t["sumL"], 4),
           "joined_frac": round(t["joins"] / t["n"], 4), "zero_row_chains_frac": round(t["zero_row"] / t["n"], 4),
           "backward_chains_frac": round(t["bchains"] / t["n"], 4),
           "alpha_states": int(t["fnodes"]), "gamma_states": int(t["bnodes"]),
           "mean_own_blocks": round(float((t["hist"] * np.arange(25)).sum() / t["n"]), 2),
           "own_blocks_hist": [int(x) for x in t["hist"]]}
    print(json.dumps(out), flush=True)
