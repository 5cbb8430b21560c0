"""An INDEPENDENT statement of the Plan7 local multihit model, for tests only: HMMER3/f text -> profile (occupancy
entry distribution, local exits, length model) -> Forward in float64 log space, node by node (no striping, no odds
ratios, no rescaling).  It shares no code with oracle/ and none of its numerics; tests compare the two at 1e-3 nats,
which checks the oracle's profile configuration and recursion, not its bit patterns.
"""
import math

import numpy as np

NEG = -1e300


def parse_hmms(text):
    """list of dict(name, M, mat[M+1][4] (prob), t[M+1][7] (prob: MM MI MD IM II DM DD), node 0 = begin)"""
    out = []
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        if not lines[i].startswith("HMMER3"):
            i += 1
            continue
        name, M = None, 0
        stats = {}
        while not lines[i].startswith("HMM "):
            if lines[i].startswith("STATS LOCAL"):
                tok = lines[i].split()
                stats[tok[2]] = (float(tok[3]), float(tok[4]))
            if lines[i].startswith("NAME"):
                name = lines[i].split()[1]
            if lines[i].startswith("LENG"):
                M = int(lines[i].split()[1])
            i += 1
        i += 2                                   # header line + transition header
        p = lambda tok: 0.0 if tok == "*" else math.exp(-float(tok))
        compo = None
        if lines[i].split()[0] == "COMPO":
            compo = [p(x) for x in lines[i].split()[1:5]]
            i += 1
        mat = np.zeros((M + 1, 4))
        t = np.zeros((M + 1, 7))
        i += 1                                   # node-0 insert emissions
        t[0] = [p(x) for x in lines[i].split()[:7]]
        i += 1
        for k in range(1, M + 1):
            tok = lines[i].split()
            assert int(tok[0]) == k
            mat[k] = [p(x) for x in tok[1:5]]
            i += 2                               # match line, insert line
            t[k] = [p(x) for x in lines[i].split()[:7]]
            i += 1
        out.append(dict(name=name, M=M, mat=mat, t=t, compo=compo, stats=stats))
    return out


def forward_nats(h, seq, L_model=None, unihit=False):
    """ln P(seq | local profile at target length L) - no null model subtracted; emissions as odds vs 0.25.
    multihit (default): nj = 1, E->J and E->C at 1/2 each; unihit: nj = 0, E->C at 1 (the mode envelopes are re-scored in)"""
    M, mat, t = h["M"], h["mat"], h["t"]
    L = len(seq)
    Lm = L if L_model is None else L_model
    MM, MI, MD, IM, II, DM, DD = range(7)
    ln = lambda x: math.log(x) if x > 0 else NEG
    # entry: match occupancy (p7_hmm_CalculateOccupancy) -> B->Mk = occ[k] / sum_k occ[k] (M-k+1)
    occ = np.zeros(M + 1)
    occ[1] = t[0][MI] + t[0][MM]
    for k in range(2, M + 1):
        occ[k] = occ[k - 1] * (t[k - 1][MM] + t[k - 1][MI]) + (1.0 - occ[k - 1]) * t[k - 1][DM]
    Z = sum(occ[k] * (M - k + 1) for k in range(1, M + 1))
    bm = [NEG] + [ln(occ[k] / Z) for k in range(1, M + 1)]
    lt = np.vectorize(ln)(t)
    nj = 0.0 if unihit else 1.0
    pmove = (2.0 + nj) / (Lm + 2.0 + nj)
    lmove, lloop = math.log(pmove), math.log(1.0 - pmove)
    lE = 0.0 if unihit else math.log(0.5)
    code = {"A": (0,), "C": (1,), "G": (2,), "T": (3,), "U": (3,), "R": (0, 2), "Y": (1, 3), "M": (0, 1), "K": (2, 3), "S": (1, 2),
            "W": (0, 3), "H": (0, 1, 3), "B": (1, 2, 3), "V": (0, 1, 2), "D": (0, 2, 3), "N": (0, 1, 2, 3)}
    sc = np.log(mat / 0.25, where=mat > 0, out=np.full(mat.shape, NEG))
    lse = np.logaddexp
    Mv = np.full(M + 1, NEG); Iv = np.full(M + 1, NEG); Dv = np.full(M + 1, NEG)
    xN, xB, xJ, xC = 0.0, lmove, NEG, NEG
    for i in range(1, L + 1):
        xs = code[seq[i - 1].upper()]
        em = sc[:, xs[0]] if len(xs) == 1 else sc[:, list(xs)].mean(axis=1)      # a degenerate symbol scores its expected score
        Mn = np.full(M + 1, NEG); In = np.full(M + 1, NEG); Dn = np.full(M + 1, NEG)
        xE = NEG
        for k in range(1, M + 1):
            s = xB + bm[k]
            if k > 1:
                s = lse(s, lse(lse(Mv[k - 1] + lt[k - 1][MM], Iv[k - 1] + lt[k - 1][IM]), Dv[k - 1] + lt[k - 1][DM]))
            Mn[k] = s + em[k]
            if k < M:
                In[k] = lse(Mv[k] + lt[k][MI], Iv[k] + lt[k][II])      # insert emission odds = 1
            if k > 1:
                Dn[k] = lse(Mn[k - 1] + lt[k - 1][MD], Dn[k - 1] + lt[k - 1][DD])
            xE = lse(xE, lse(Mn[k], Dn[k]))                             # local exits cost nothing
        xJ = NEG if unihit else lse(xJ + lloop, xE + lE)
        xC = lse(xC + lloop, xE + lE)
        xN = xN + lloop
        xB = lse(xN + lmove, xJ + lmove)
        Mv, Iv, Dv = Mn, In, Dn
    return xC + lmove


def viterbi_filter_nats(h, seq):
    """The Viterbi filter's score in float64: the best single path of the multihit local model, exits from match states only,
    N/C/J loops at no cost and a flat -3 nats for them at the end (HMMER's ViterbiFilter approximation); no null model."""
    M, mat, t = h["M"], h["mat"], h["t"]
    L = len(seq)
    MM, MI, MD, IM, II, DM, DD = range(7)
    ln = lambda x: math.log(x) if x > 0 else NEG
    occ = np.zeros(M + 1)
    occ[1] = t[0][MI] + t[0][MM]
    for k in range(2, M + 1):
        occ[k] = occ[k - 1] * (t[k - 1][MM] + t[k - 1][MI]) + (1.0 - occ[k - 1]) * t[k - 1][DM]
    Z = sum(occ[k] * (M - k + 1) for k in range(1, M + 1))
    bm = [NEG] + [ln(occ[k] / Z) for k in range(1, M + 1)]
    lt = np.vectorize(ln)(t)
    lmove = math.log(3.0 / (L + 3.0))
    lE = math.log(0.5)
    code = {"A": (0,), "C": (1,), "G": (2,), "T": (3,), "U": (3,), "R": (0, 2), "Y": (1, 3), "M": (0, 1), "K": (2, 3), "S": (1, 2),
            "W": (0, 3), "H": (0, 1, 3), "B": (1, 2, 3), "V": (0, 1, 2), "D": (0, 2, 3), "N": (0, 1, 2, 3)}
    sc = np.log(mat / 0.25, where=mat > 0, out=np.full(mat.shape, NEG))
    Mv = np.full(M + 1, NEG); Iv = np.full(M + 1, NEG); Dv = np.full(M + 1, NEG)
    xN, xB, xJ, xC = 0.0, lmove, NEG, NEG
    for i in range(1, L + 1):
        xs = code[seq[i - 1].upper()]
        em = sc[:, xs[0]] if len(xs) == 1 else sc[:, list(xs)].mean(axis=1)
        Mn = np.full(M + 1, NEG); In = np.full(M + 1, NEG); Dn = np.full(M + 1, NEG)
        xE = NEG
        for k in range(1, M + 1):
            s = xB + bm[k]
            if k > 1:
                s = max(s, Mv[k - 1] + lt[k - 1][MM], Iv[k - 1] + lt[k - 1][IM], Dv[k - 1] + lt[k - 1][DM])
            Mn[k] = s + em[k]
            if k < M:
                In[k] = max(Mv[k] + lt[k][MI], Iv[k] + lt[k][II])
            if k > 1:
                Dn[k] = max(Mn[k - 1] + lt[k - 1][MD], Dn[k - 1] + lt[k - 1][DD])
            xE = max(xE, Mn[k])
        xC = max(xC, xE + lE)
        xJ = max(xJ, xE + lE)
        xB = max(xN + lmove, xJ + lmove)
        Mv, Iv, Dv = Mn, In, Dn
    return xC + lmove - 3.0


def decode_regions(h, seq):
    """Posterior decoding of begin / end / occupancy (p7_DomainDecoding) and the region scan of
    p7_domaindef_ByPosteriorHeuristics (rt1 0.25, rt2 0.10), all in float64 log space on the multihit model.
    Returns [(i1, i2), ...] (1-based, inclusive)."""
    M, mat, t = h["M"], h["mat"], h["t"]
    L = len(seq)
    MM, MI, MD, IM, II, DM, DD = range(7)
    ln = lambda x: math.log(x) if x > 0 else NEG
    occ = np.zeros(M + 1)
    occ[1] = t[0][MI] + t[0][MM]
    for k in range(2, M + 1):
        occ[k] = occ[k - 1] * (t[k - 1][MM] + t[k - 1][MI]) + (1.0 - occ[k - 1]) * t[k - 1][DM]
    Z = sum(occ[k] * (M - k + 1) for k in range(1, M + 1))
    bm = np.array([NEG] + [ln(occ[k] / Z) for k in range(1, M + 1)])
    lt = np.vectorize(ln)(t)
    pmove = 3.0 / (L + 3.0)
    lmove, lloop, lE = math.log(pmove), math.log(1.0 - pmove), math.log(0.5)
    code = {"A": (0,), "C": (1,), "G": (2,), "T": (3,), "U": (3,), "R": (0, 2), "Y": (1, 3), "M": (0, 1), "K": (2, 3), "S": (1, 2),
            "W": (0, 3), "H": (0, 1, 3), "B": (1, 2, 3), "V": (0, 1, 2), "D": (0, 2, 3), "N": (0, 1, 2, 3)}
    sc = np.log(mat / 0.25, where=mat > 0, out=np.full(mat.shape, NEG))
    em = np.full((L + 1, M + 1), NEG)
    for i in range(1, L + 1):
        xs = code[seq[i - 1].upper()]
        em[i] = sc[:, xs[0]] if len(xs) == 1 else sc[:, list(xs)].mean(axis=1)
    lse = np.logaddexp
    # ---- Forward, keeping the special states of every row
    fN = np.full(L + 1, NEG); fB = np.full(L + 1, NEG); fE = np.full(L + 1, NEG); fJ = np.full(L + 1, NEG); fC = np.full(L + 1, NEG)
    fN[0] = 0.0; fB[0] = lmove
    Mv = np.full(M + 1, NEG); Iv = np.full(M + 1, NEG); Dv = np.full(M + 1, NEG)
    for i in range(1, L + 1):
        Mn = np.full(M + 1, NEG); In = np.full(M + 1, NEG); Dn = np.full(M + 1, NEG)
        xE = NEG
        for k in range(1, M + 1):
            s = fB[i - 1] + bm[k]
            if k > 1:
                s = lse(s, lse(lse(Mv[k - 1] + lt[k - 1][MM], Iv[k - 1] + lt[k - 1][IM]), Dv[k - 1] + lt[k - 1][DM]))
            Mn[k] = s + em[i][k]
            if k < M:
                In[k] = lse(Mv[k] + lt[k][MI], Iv[k] + lt[k][II])
            if k > 1:
                Dn[k] = lse(Mn[k - 1] + lt[k - 1][MD], Dn[k - 1] + lt[k - 1][DD])
            xE = lse(xE, lse(Mn[k], Dn[k]))
        fE[i] = xE
        fJ[i] = lse(fJ[i - 1] + lloop, xE + lE)
        fC[i] = lse(fC[i - 1] + lloop, xE + lE)
        fN[i] = fN[i - 1] + lloop
        fB[i] = lse(fN[i] + lmove, fJ[i] + lmove)
        Mv, Iv, Dv = Mn, In, Dn
    total = fC[L] + lmove
    # ---- Backward (b*[i] = ln P(x_{i+1..L} | state at i))
    bN = np.full(L + 1, NEG); bB = np.full(L + 1, NEG); bE = np.full(L + 1, NEG); bJ = np.full(L + 1, NEG); bC = np.full(L + 1, NEG)
    bC[L] = lmove
    bE[L] = bC[L] + lE
    Mb = np.full(M + 2, NEG); Ib = np.full(M + 2, NEG); Db = np.full(M + 2, NEG)
    for k in range(M, 0, -1):                          # row L: exit directly, or through the delete states to the right
        Mb[k] = bE[L] if k == M else lse(bE[L], lt[k][MD] + Db[k + 1])
        Db[k] = bE[L] if k == M else lse(bE[L], lt[k][DD] + Db[k + 1])
    for i in range(L - 1, -1, -1):
        # B at row i enters a match state that emits residue i+1
        xB = NEG
        for k in range(1, M + 1):
            xB = lse(xB, bm[k] + em[i + 1][k] + Mb[k])
        bB[i] = xB
        bJ[i] = lse(bJ[i + 1] + lloop, xB + lmove)
        bC[i] = bC[i + 1] + lloop
        bE[i] = lse(bJ[i] + lE, bC[i] + lE)
        bN[i] = lse(bN[i + 1] + lloop, xB + lmove)
        if i == 0:
            break
        Mn = np.full(M + 2, NEG); In = np.full(M + 2, NEG); Dn = np.full(M + 2, NEG)
        for k in range(M, 0, -1):
            nm = em[i + 1][k + 1] + Mb[k + 1] if k < M else NEG           # next row's match state k+1 (emits residue i+1)
            Mn[k] = bE[i]
            Dn[k] = bE[i]
            if k < M:
                Mn[k] = lse(Mn[k], lse(lse(lt[k][MM] + nm, lt[k][MI] + Ib[k]), lt[k][MD] + Dn[k + 1]))
                In[k] = lse(lt[k][IM] + nm, lt[k][II] + Ib[k])
                Dn[k] = lse(Dn[k], lse(lt[k][DM] + nm, lt[k][DD] + Dn[k + 1]))
        Mb, Ib, Db = Mn, In, Dn
    assert abs((bN[0]) - total) < 1e-6 * max(1.0, abs(total)), (bN[0], total)
    # ---- decoding + region scan
    btot = np.zeros(L + 1); etot = np.zeros(L + 1); mocc = np.zeros(L + 1)
    for i in range(1, L + 1):
        btot[i] = btot[i - 1] + math.exp(fB[i - 1] + bB[i - 1] - total)
        etot[i] = etot[i - 1] + math.exp(fE[i] + bE[i] - total)
        njcp = math.exp(fN[i - 1] + bN[i] + lloop - total) + math.exp(fJ[i - 1] + bJ[i] + lloop - total) + math.exp(fC[i - 1] + bC[i] + lloop - total)
        mocc[i] = 1.0 - njcp
    regions = []
    i1, trig = -1, False
    for i in range(1, L + 1):
        if not trig:
            if mocc[i] - (btot[i] - btot[i - 1]) < 0.10:
                i1 = i
            elif i1 == -1:
                i1 = i
            if mocc[i] >= 0.25:
                trig = True
        elif mocc[i] - (etot[i] - etot[i - 1]) < 0.10:
            regions.append((i1, i))
            i1, trig = -1, False
    return regions


def domain_bits(h, seq, ienv, jenv):
    """Bit score of one envelope as hmmsearch reports it (column 14 of --domtblout), independently in float64:
    unihit Forward over the envelope with the length model at the full target length, + the flanks' N/C loops,
    - null1, - the null2 bias correction (p7_Null2_ByExpectation: expected state usage from posterior decoding of the
    envelope; prior omega = 1/256).  Returns (bits, domcorrection_nats, envsc_nats)."""
    M, mat, t = h["M"], h["mat"], h["t"]
    L = len(seq)
    sub = seq[ienv - 1:jenv]
    Ld = len(sub)
    MM, MI, MD, IM, II, DM, DD = range(7)
    ln = lambda x: math.log(x) if x > 0 else NEG
    occ = np.zeros(M + 1)
    occ[1] = t[0][MI] + t[0][MM]
    for k in range(2, M + 1):
        occ[k] = occ[k - 1] * (t[k - 1][MM] + t[k - 1][MI]) + (1.0 - occ[k - 1]) * t[k - 1][DM]
    Z = sum(occ[k] * (M - k + 1) for k in range(1, M + 1))
    bm = np.array([NEG] + [ln(occ[k] / Z) for k in range(1, M + 1)])
    lt = np.vectorize(ln)(t)
    pmove = 2.0 / (L + 2.0)                                  # unihit, length model stays at L
    lmove, lloop = math.log(pmove), math.log(1.0 - pmove)
    code = {"A": (0,), "C": (1,), "G": (2,), "T": (3,), "U": (3,), "R": (0, 2), "Y": (1, 3), "M": (0, 1), "K": (2, 3), "S": (1, 2),
            "W": (0, 3), "H": (0, 1, 3), "B": (1, 2, 3), "V": (0, 1, 2), "D": (0, 2, 3), "N": (0, 1, 2, 3)}
    odds = mat / 0.25
    sc = np.log(odds, where=mat > 0, out=np.full(mat.shape, NEG))
    em = np.full((Ld + 1, M + 1), NEG)
    for i in range(1, Ld + 1):
        xs = code[sub[i - 1].upper()]
        em[i] = sc[:, xs[0]] if len(xs) == 1 else sc[:, list(xs)].mean(axis=1)
    lse = np.logaddexp
    fM = np.full((Ld + 1, M + 2), NEG); fI = np.full((Ld + 1, M + 2), NEG); fD = np.full((Ld + 1, M + 2), NEG)
    fN = np.full(Ld + 1, NEG); fB = np.full(Ld + 1, NEG); fE = np.full(Ld + 1, NEG); fC = np.full(Ld + 1, NEG)
    fN[0] = 0.0; fB[0] = lmove
    for i in range(1, Ld + 1):
        xE = NEG
        for k in range(1, M + 1):
            s = fB[i - 1] + bm[k]
            if k > 1:
                s = lse(s, lse(lse(fM[i - 1][k - 1] + lt[k - 1][MM], fI[i - 1][k - 1] + lt[k - 1][IM]), fD[i - 1][k - 1] + lt[k - 1][DM]))
            fM[i][k] = s + em[i][k]
            if k < M:
                fI[i][k] = lse(fM[i - 1][k] + lt[k][MI], fI[i - 1][k] + lt[k][II])
            if k > 1:
                fD[i][k] = lse(fM[i][k - 1] + lt[k - 1][MD], fD[i][k - 1] + lt[k - 1][DD])
            xE = lse(xE, lse(fM[i][k], fD[i][k]))
        fE[i] = xE
        fC[i] = lse(fC[i - 1] + lloop, xE)                   # unihit: E->C with probability 1
        fN[i] = fN[i - 1] + lloop
        fB[i] = fN[i] + lmove
    envsc = fC[Ld] + lmove
    bM = np.full((Ld + 2, M + 2), NEG); bI = np.full((Ld + 2, M + 2), NEG); bD = np.full((Ld + 2, M + 2), NEG)
    bN = np.full(Ld + 2, NEG); bC = np.full(Ld + 2, NEG); bE = np.full(Ld + 2, NEG); bB = np.full(Ld + 2, NEG)
    bC[Ld] = lmove
    bE[Ld] = bC[Ld]
    for k in range(M, 0, -1):
        bM[Ld][k] = bE[Ld] if k == M else lse(bE[Ld], lt[k][MD] + bD[Ld][k + 1])
        bD[Ld][k] = bE[Ld] if k == M else lse(bE[Ld], lt[k][DD] + bD[Ld][k + 1])
    for i in range(Ld - 1, -1, -1):
        xB = NEG
        for k in range(1, M + 1):
            xB = lse(xB, bm[k] + em[i + 1][k] + bM[i + 1][k])
        bB[i] = xB
        bC[i] = bC[i + 1] + lloop
        bE[i] = bC[i]
        bN[i] = lse(bN[i + 1] + lloop, xB + lmove)
        if i == 0:
            break
        for k in range(M, 0, -1):
            nm = em[i + 1][k + 1] + bM[i + 1][k + 1] if k < M else NEG
            bM[i][k] = bE[i]
            bD[i][k] = bE[i]
            if k < M:
                bM[i][k] = lse(bM[i][k], lse(lse(lt[k][MM] + nm, lt[k][MI] + bI[i + 1][k]), lt[k][MD] + bD[i][k + 1]))
                bI[i][k] = lse(lt[k][IM] + nm, lt[k][II] + bI[i + 1][k])
                bD[i][k] = lse(bD[i][k], lse(lt[k][DM] + nm, lt[k][DD] + bD[i][k + 1]))
    # expected usage of the emitting states over the envelope
    ppM = np.zeros(M + 1); ppI = 0.0; ppX = 0.0
    for i in range(1, Ld + 1):
        for k in range(1, M + 1):
            ppM[k] += math.exp(fM[i][k] + bM[i][k] - envsc)
            if k < M:
                ppI += math.exp(fI[i][k] + bI[i][k] - envsc)
        ppX += math.exp(fN[i - 1] + lloop + bN[i] - envsc) + math.exp(fC[i - 1] + lloop + bC[i] - envsc)
    null2 = np.array([(sum(ppM[k] * odds[k][x] for k in range(1, M + 1)) + ppI + ppX) / Ld for x in range(4)])
    domcorr = 0.0
    for ch in sub:
        xs = code[ch.upper()]
        domcorr += math.log(null2[xs[0]]) if len(xs) == 1 else math.log(np.mean([null2[x] for x in xs]))
    nullsc = L * math.log(L / (L + 1.0)) + math.log(1.0 / (L + 1.0))
    dombias = math.log(1.0 + math.exp(math.log(1.0 / 256.0) + domcorr))
    bits = (envsc + (L - Ld) * math.log(L / (L + 3.0)) - (nullsc + dombias)) / math.log(2.0)
    return bits, domcorr, envsc


def msv_nats(h, seq):
    """Generic MSV score (p7_GMSV): best set of ungapped local diagonals, uniform entry 2/(M(M+1)), multihit."""
    M, mat = h["M"], h["mat"]
    L = len(seq)
    tloop, tmove = math.log(L / (L + 3.0)), math.log(3.0 / (L + 3.0))
    tbm, te = math.log(2.0 / (M * (M + 1.0))), math.log(0.5)
    code = {"A": (0,), "C": (1,), "G": (2,), "T": (3,), "U": (3,), "R": (0, 2), "Y": (1, 3), "M": (0, 1), "K": (2, 3), "S": (1, 2),
            "W": (0, 3), "H": (0, 1, 3), "B": (1, 2, 3), "V": (0, 1, 2), "D": (0, 2, 3), "N": (0, 1, 2, 3)}
    sc = np.log(mat / 0.25, where=mat > 0, out=np.full(mat.shape, NEG))
    prev = np.full(M + 1, NEG)
    xN, xB, xJ, xC = 0.0, tmove, NEG, NEG
    for ch in seq:
        xs = code[ch.upper()]
        em = sc[:, xs[0]] if len(xs) == 1 else sc[:, list(xs)].mean(axis=1)
        cur = np.full(M + 1, NEG)
        cur[1:] = em[1:] + np.maximum(prev[:-1], xB + tbm)
        xE = cur[1:].max()
        xJ = max(xJ + tloop, xE + te)
        xC = max(xC + tloop, xE + te)
        xN = xN + tloop
        xB = max(xN + tmove, xJ + tmove)
        prev = cur
    return xC + tmove


def bias_filter_nats(h, seq):
    """p7_bg_FilterScore: ln P(seq | 2-state composition HMM) as odds against the iid background, plus the null length
    model; state 0 = background (mean run 400, start 0.999, re-set to the target length), state 1 = the model's COMPO
    (mean run M/8)."""
    L = len(seq)
    p1 = L / (L + 1.0)
    L1 = h["M"] / 8.0
    t = [[p1, 1.0 - p1], [1.0 / (L1 + 1.0), L1 / (L1 + 1.0)]]
    sets = {"A": (0,), "C": (1,), "G": (2,), "T": (3,), "U": (3,), "R": (0, 2), "Y": (1, 3), "M": (0, 1), "K": (2, 3), "S": (1, 2),
            "W": (0, 3), "H": (0, 1, 3), "B": (1, 2, 3), "V": (0, 1, 2), "D": (0, 2, 3), "N": (0, 1, 2, 3)}
    e = [[0.25] * 4, list(h["compo"])]
    def eo(k, ch):
        xs = sets[ch.upper()]
        return sum(e[k][x] for x in xs) / sum(0.25 for _ in xs)
    a = [0.999 * eo(0, seq[0]), 0.001 * eo(1, seq[0])]
    logsc = 0.0
    for ch in seq[1:]:
        m = max(a)
        logsc += math.log(m)
        a = [a[0] / m, a[1] / m]
        a = [(a[0] * t[0][0] + a[1] * t[1][0]) * eo(0, ch), (a[0] * t[0][1] + a[1] * t[1][1]) * eo(1, ch)]
    logsc += math.log(a[0] + a[1])
    return logsc + L * math.log(p1) + math.log(1.0 - p1)
