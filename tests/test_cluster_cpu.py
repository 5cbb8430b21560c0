"""CPU tests for the clustering oracle (oracle/orc_cluster.c; row a2, vsearch --cluster_size restated).

PARITY UNPINNED against a real vsearch (the reference holds no fixture for this path): these tests pin the C
restatement to (a) hand-computable answers and (b) an independent pure-Python statement of the same procedure,
so that the thing the GPU engine is compared with (tests/test_gpu_cluster.py) is itself checked twice.
"""
import numpy as np

import orc

_RC = str.maketrans("ACGTN", "TGCAN")
_IUPAC = {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "R": 5, "Y": 10, "M": 3, "K": 12, "S": 6, "W": 9, "H": 11, "B": 14,
          "V": 7, "D": 13, "N": 15}


# ---------------------------------------------------------------- an independent statement of the procedure
def py_align(q, t):
    """global alignment on (score, matches, -columns) tuples; terminal gaps 2+k uncounted, interior 20+2k counted"""
    NEG = (-10 ** 9, 0, 0)
    add = lambda a, b: (a[0] + b[0], a[1] + b[1], a[2] + b[2])
    Lq, Lt = len(q), len(t)
    H = [[NEG] * (Lt + 1) for _ in range(Lq + 1)]
    E = [[NEG] * (Lt + 1) for _ in range(Lq + 1)]
    F = [[NEG] * (Lt + 1) for _ in range(Lq + 1)]
    for i in range(Lq + 1):
        te = i in (0, Lq)
        goE, geE = ((-3, 0, 0), (-1, 0, 0)) if te else ((-22, 0, -1), (-2, 0, -1))
        for j in range(Lt + 1):
            tf = j in (0, Lt)
            goF, geF = ((-3, 0, 0), (-1, 0, 0)) if tf else ((-22, 0, -1), (-2, 0, -1))
            if i == 0 and j == 0:
                H[0][0] = (0, 0, 0)
                continue
            best = NEG
            if j > 0:
                E[i][j] = max(add(H[i][j - 1], goE), add(E[i][j - 1], geE))
                best = max(best, E[i][j])
            if i > 0:
                F[i][j] = max(add(H[i - 1][j], goF), add(F[i - 1][j], geF))
                best = max(best, F[i][j])
            if i > 0 and j > 0:
                a, b = _IUPAC[q[i - 1]], _IUPAC[t[j - 1]]
                una = a in (1, 2, 4, 8) and b in (1, 2, 4, 8)
                d = ((2, 1, -1) if a == b else (-4, 0, -1)) if una else ((0, 1, -1) if a & b else (0, 0, -1))
                best = max(best, add(H[i - 1][j - 1], d))
            H[i][j] = best
    s, m, c = H[Lq][Lt]
    return s, m, -c


def py_dust(seq):
    """vsearch's DUST soft mask (mask.cc dust()/wo(), after Tatusov & Lipman), stated independently of orc_dust: windows of 64
    that advance by 32, 3-mer repeat score 10 * sum / j, masked above 20, the first best interval in (i, j) order"""
    code = {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3}
    s = [code.get(c, 0) for c in seq.upper()]
    masked = [False] * len(s)
    i = 0
    while i < len(s):
        win = s[i:i + 64]
        n = len(win)
        best = (0, 0, 0)
        if n - 7 >= 0:
            tri = [((win[j - 2] if j >= 2 else 0) << 4 | (win[j - 1] if j >= 1 else 0) << 2 | win[j]) for j in range(n)]
            for a in range(n - 7):
                seen, total = {}, 0
                for j in range(2, n - a):
                    w = tri[a + j]
                    c = seen.get(w, 0)
                    if c:
                        total += c
                        v = 10 * total // j
                        if v > best[0]:
                            best = (v, a, j)
                    seen[w] = c + 1
        v, a, j = best
        if v > 20:
            for k in range(a + i, a + j + i + 1):
                masked[k] = True
            if a + j < 32:
                i += 32 - (a + j)
        i += 32
    return masked


def py_words(s, masked=None):
    ok = [ch in "ACGT" and not (masked and masked[i]) for i, ch in enumerate(s)]
    return {s[i:i + 8] for i in range(len(s) - 7) if all(ok[i:i + 8])}


def py_cluster(reads, names, cid, strand_both=True, minlen=32, dust=True):
    masks = [py_dust(r) if dust else None for r in reads]
    order = sorted((i for i in range(len(reads)) if len(reads[i]) >= minlen), key=lambda i: (names[i].encode(), i))
    cents = []                                      # (read index, words, position)
    rep_of = [-1] * len(reads)
    strand = [1] * len(reads)
    pct = [-1.0] * len(reads)
    for pos, r in enumerate(order):
        best = None
        for s in (0, 1) if strand_both else (0,):
            q = reads[r] if s == 0 else reads[r][::-1].translate(_RC)
            qw = py_words(q, masks[r] if s == 0 or masks[r] is None else masks[r][::-1])
            if not qw:
                continue
            minm = min(12, len(qw))
            cands = []
            for ci, (cr, cw, cpos) in enumerate(cents):
                n = len(qw & cw)
                if n >= minm:
                    cands.append((-n, len(reads[cr]), cpos, ci))
            rejects = 0
            for _, _, _, ci in sorted(cands):
                if rejects >= 32:
                    break
                sc, m, cols = py_align(q, reads[cents[ci][0]])
                pid = 100.0 * m / cols if cols > 0 else 0.0
                if pid >= 100.0 * cid:
                    if best is None or pid > best[1]:
                        best = (ci, pid, 1 if s == 0 else -1)
                    break
                rejects += 1
        if best is None:
            rep_of[r] = r
            cents.append((r, py_words(reads[r], masks[r]), pos))
        else:
            rep_of[r] = cents[best[0]][0]
            strand[r] = best[2]
            pct[r] = best[1]
    return rep_of, strand, pct, order


# ---------------------------------------------------------------- tests
def test_alignment_known_answers():
    assert orc.align_identity("ACGTACGTAC", "ACGTACGTAC") == (20, 10, 10)
    assert orc.align_identity("ACGTACGTAC", "ACGTTCGTAC") == (14, 9, 10)
    # overhangs are terminal gaps: cost 2 + k each, not counted in the identity
    assert orc.align_identity("ACGTACGTACGGGTTT", "TACGTACGGG") == (10, 10, 10)
    rng = np.random.default_rng(3)
    t = "".join(rng.choice(list("ACGT"), 300))
    assert orc.align_identity(t[:150] + t[151:], t) == (2 * 299 - 22, 299, 300)          # one interior gap column
    assert orc.align_identity(t[:100] + "N" + t[101:], t) == (2 * 299, 300, 300)          # N: score 0, a match
    assert orc.align_identity(t[:100] + "R" + t[101:], t)[1] in (299, 300)                # R matches A/G only
    assert orc.align_identity(t[5:], t) == (2 * 295 - 7, 295, 295)                        # truncated copy: 100 %


def test_alignment_matches_the_tuple_dp():
    rng = np.random.default_rng(4)
    for _ in range(60):
        L = int(rng.integers(8, 40))
        a = "".join(rng.choice(list("ACGT"), L))
        b = list(a)
        for _ in range(int(rng.integers(0, 6))):
            k = int(rng.integers(0, len(b)))
            op = rng.random()
            if op < 0.3 and len(b) > 4:
                del b[k]
            elif op < 0.6:
                b.insert(k, str(rng.choice(list("ACGTN"))))
            else:
                b[k] = str(rng.choice(list("ACGTNRY")))
        b = "".join(b)
        assert orc.align_identity(a, b) == py_align(a, b), (a, b)
        assert orc.align_identity(b, a) == py_align(b, a), (b, a)


def _small_library(seed, n, n_tmpl, L):
    rng = np.random.default_rng(seed)
    flank = "".join(rng.choice(list("ACGT"), 16))
    tmpl = [flank + "".join(rng.choice(list("ACGT"), L - 16)) for _ in range(n_tmpl)]
    reads, names = [], []
    for i in range(n):
        s = list(tmpl[int(rng.integers(0, n_tmpl))])
        for _ in range(int(rng.integers(0, 4))):
            k = int(rng.integers(0, len(s)))
            s[k] = str(rng.choice(list("ACGTN")))
        if rng.random() < 0.2:
            del s[int(rng.integers(0, len(s)))]
        s = "".join(s)
        if rng.random() < 0.3:
            s = s[::-1].translate(_RC)
        reads.append(s)
        names.append("x%04d" % int(rng.integers(0, 5000)))
    return reads, names


def test_cluster_matches_the_python_statement():
    for seed, cid in ((1, 0.97), (2, 0.95), (3, 0.99)):
        reads, names = _small_library(seed, 70, 6, 56)
        codes, off = orc.digitize(reads)
        o = orc.cluster(codes, off, names, cid)
        rep_of, strand, pct, order = py_cluster(reads, names, cid)
        assert o["order"].tolist() == order
        assert o["rep_of"].tolist() == rep_of
        assert o["strand"].tolist() == strand
        assert o["pct_id"].tolist() == pct
        assert 6 <= o["n_centroids"] < 70


def test_cluster_semantics():
    rng = np.random.default_rng(9)
    t = "".join(rng.choice(list("ACGT"), 100))
    one = t[:50] + ("A" if t[50] != "A" else "C") + t[51:]
    # label order decides who becomes the centroid, not input order
    codes, off = orc.digitize([one, t])
    o = orc.cluster(codes, off, ["b", "a"], 0.99)
    assert o["order"].tolist() == [1, 0] and o["rep_of"].tolist() == [1, 1] and o["pct_id"][0] == 99.0
    # the threshold is inclusive and exact
    assert orc.cluster(codes, off, ["b", "a"], 0.9901)["rep_of"].tolist() == [0, 1]
    # reverse complements join on the minus strand; plus-only keeps them apart
    codes, off = orc.digitize([t, t[::-1].translate(_RC)])
    assert orc.cluster(codes, off, ["a", "b"], 0.99)["strand"].tolist() == [1, -1]
    assert orc.cluster(codes, off, ["a", "b"], 0.99, strand_both=False)["rep_of"].tolist() == [0, 1]
    # no shared words, no candidate: identical reads whose every 8-mer holds an N stay apart
    holes = "".join("N" if i % 7 == 0 else c for i, c in enumerate(t))
    codes, off = orc.digitize([holes, holes])
    assert orc.cluster(codes, off, ["a", "b"], 0.9)["rep_of"].tolist() == [0, 1]
    # reads below --minseqlength 32 are dropped
    codes, off = orc.digitize([t, t[:31]])
    assert orc.cluster(codes, off, ["a", "b"], 0.99)["rep_of"].tolist() == [0, -1]
    # without labels the input order is the processing order
    codes, off = orc.digitize([one, t])
    assert orc.cluster(codes, off, None, 0.99)["rep_of"].tolist() == [0, 0]


def _low_complexity_library(seed, n):
    """templates with homopolymer / microsatellite stretches: DUST removes their words from the seeds of queries and centroids"""
    rng = np.random.default_rng(seed)
    rnd = lambda k: "".join(rng.choice(list("ACGT"), k))
    tmpl = []
    for t in range(5):
        core = [rnd(30), "A" * int(rng.integers(12, 30)), rnd(25), "CA" * int(rng.integers(8, 20)), rnd(20), "TTG" * int(rng.integers(6, 12)), rnd(30)]
        tmpl.append("".join(core[k] for k in rng.permutation(len(core))))
    reads, names = [], []
    for i in range(n):
        s = list(tmpl[int(rng.integers(0, 5))])
        for _ in range(int(rng.integers(0, 4))):
            s[int(rng.integers(0, len(s)))] = str(rng.choice(list("ACGTN")))
        s = "".join(s)
        if rng.random() < 0.3:
            s = s[::-1].translate(_RC)
        reads.append(s)
        names.append("d%04d" % int(rng.integers(0, 5000)))
    return reads, names


def test_dust_mask_matches_the_python_statement():
    rng = np.random.default_rng(4)
    rnd = lambda k: "".join(rng.choice(list("ACGT"), k))
    seqs = [rnd(200), rnd(80) + "A" * 40 + rnd(80), rnd(60) + "AC" * 25 + rnd(90), rnd(50) + "ACG" * 12 + rnd(150), "T" * 30, "ACGTTGCA" * 20,
            rnd(5), rnd(7), rnd(8), "A" * 8, rnd(63) + "G" * 20, rnd(33) + "N" * 30 + rnd(40), rnd(31) + "GA" * 9 + rnd(28) + "C" * 11 + rnd(70)]
    reads, _ = _low_complexity_library(7, 40)
    n_masked = 0
    for s in seqs + reads:
        m = orc.dust(s)
        assert m.tolist() == py_dust(s), s
        n_masked += int(m.sum())
    assert not orc.dust(seqs[0]).any() and orc.dust(seqs[1])[85:115].all() and orc.dust("T" * 30).all() and n_masked > 1000
    # anything but A C G T counts as A: a run of N is a homopolymer to DUST
    assert orc.dust(seqs[11])[40:55].all()


def test_cluster_with_dust_masking_matches_the_python_statement(monkeypatch):
    """vsearch's default --qmask dust: words that touch a soft-masked symbol are left out of the k-mer sets of queries AND
    centroids; the alignment still sees every symbol.  C restatement == the Python statement with the masking on (default)
    and off (ORC_QMASK=none), and the masking does change outcomes on low-complexity reads."""
    reads, names = _low_complexity_library(11, 80)
    codes, off = orc.digitize(reads)
    res = {}
    for dust in (True, False):
        if not dust:
            monkeypatch.setenv("ORC_QMASK", "none")
        o = orc.cluster(codes, off, names, 0.97)
        rep_of, strand, pct, order = py_cluster(reads, names, 0.97, dust=dust)
        assert o["order"].tolist() == order and o["rep_of"].tolist() == rep_of and o["strand"].tolist() == strand and o["pct_id"].tolist() == pct
        res[dust] = (o["rep_of"].tolist(), o["n_alignments"])
    assert res[True] != res[False]              # fewer shared words -> other candidates, other walks
