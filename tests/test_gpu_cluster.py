"""GPU parity tests for row a2 (greedy clustering, vsearch --cluster_size restated): the HIP engine through the
C ABI against oracle/orc_cluster.c.  Bar: identical cluster maps, strands, processing order and identities
(the identity is a ratio of two integers: compared as exact doubles).  The speculative-window scheme must give
the sequential answer for every window size.
"""
import os

import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu

_RC = str.maketrans("ACGTN", "TGCAN")


def _noisy_library(seed, n_reads, n_tmpl, length, max_err=6, indel=True, n_rate=0.002, rc_rate=0.3, shared_flank=60):
    """templates share a conserved flank (like 5.8S/LSU in amplicons); reads carry substitutions, a few indels, N"""
    rng = np.random.default_rng(seed)
    acgt = np.array(list("ACGT"))
    flank_l = "".join(acgt[rng.integers(0, 4, shared_flank)])
    flank_r = "".join(acgt[rng.integers(0, 4, shared_flank)])
    lens = rng.integers(length[0], length[1] + 1, n_tmpl)
    tmpl = [flank_l + "".join(acgt[rng.integers(0, 4, int(L) - 2 * shared_flank)]) + flank_r for L in lens]
    # a few templates that are near copies of others (clusters that compete)
    for t in range(1, n_tmpl, 5):
        s = list(tmpl[t - 1])
        for _ in range(int(rng.integers(1, 8))):
            s[int(rng.integers(0, len(s)))] = str(acgt[rng.integers(0, 4)])
        tmpl[t] = "".join(s)
    reads, names = [], []
    for i in range(n_reads):
        s = list(tmpl[int(rng.integers(0, n_tmpl))])
        for _ in range(int(rng.integers(0, max_err + 1)) if rng.random() < 0.6 else 0):
            k = int(rng.integers(0, len(s)))
            op = rng.random()
            if indel and op < 0.15:
                del s[k]
            elif indel and op < 0.3:
                s.insert(k, str(acgt[rng.integers(0, 4)]))
            else:
                s[k] = str(acgt[rng.integers(0, 4)])
        if rng.random() < 0.1:                       # truncated copies: terminal gaps are not counted
            cut = int(rng.integers(1, 12))
            s = s[cut:] if rng.random() < 0.5 else s[:-cut]
        s = "".join(s)
        if n_rate:
            s = "".join("N" if rng.random() < n_rate else c for c in s)
        if rng.random() < rc_rate:
            s = s[::-1].translate(_RC)
        reads.append(s)
        names.append("M0:%06d:%04d" % (int(rng.integers(0, 10 ** 6)), i % 977))
    return reads, names


def _compare(engine, reads, names, cid, strand_both=True):
    engine.set_reads(reads, names)
    engine.cluster(cid, strand_both=strand_both)
    rep_of, strand, uniq_of = engine.get_derep()
    pct, order = engine.get_cluster()
    codes, off = orc.digitize(reads)
    o = orc.cluster(codes, off, names, cid, strand_both=strand_both)
    assert np.array_equal(order, o["order"])
    assert np.array_equal(rep_of, o["rep_of"])
    assert np.array_equal(strand, o["strand"])
    assert np.array_equal(pct.view(np.uint64), o["pct_id"].view(np.uint64))
    st = engine.stats()
    assert st["n_unique"] == o["n_centroids"]
    # the engine may redo alignments (speculation) and may skip ones that cannot change the outcome (minus strand when the
    # plus strand holds a 100 % hit), so the counts are not comparable; the outcomes above are
    assert st["cl_alignments"] > 0 or o["n_alignments"] == 0
    return o, st


@pytest.mark.parametrize("cid", [0.97, 0.99, 0.995])
def test_cluster_matches_oracle(engine, cid):
    reads, names = _noisy_library(11, 3000, 60, (280, 310))
    o, st = _compare(engine, reads, names, cid)
    assert 60 <= o["n_centroids"] < 3000 and (o["strand"] < 0).sum() > 100


def test_cluster_candidate_lists_grow_when_they_overflow(engine, monkeypatch):
    # a strand's candidate list holds what one chunk of centroids can append (4096 keys); forced tiny here: the window is
    # searched again with lists four times the size until nothing overflows, and the outcome is the oracle's all the same
    monkeypatch.setenv("ITSX_CL_CCAP", "40")
    reads, names = _noisy_library(41, 1500, 30, (200, 260), n_rate=0.01)
    _compare(engine, reads, names, 0.985)


@pytest.mark.parametrize("window", ["1", "7", "64", "4096"])
def test_cluster_is_independent_of_the_window(engine, window, monkeypatch):
    monkeypatch.setenv("ITSX_CL_WINDOW", window)
    reads, names = _noisy_library(12, 700, 25, (120, 160), shared_flank=30)
    _compare(engine, reads, names, 0.98)


@pytest.mark.parametrize("window", ["16", "4096"])
def test_cluster_many_ambiguous_symbols(engine, window, monkeypatch):
    # N matches anything: 100 % hits with fewer shared words than a worse-matching newcomer -- the walk replay,
    # the reject budget and the 100 % plus-strand shortcut all get exercised
    monkeypatch.setenv("ITSX_CL_WINDOW", window)
    reads, names = _noisy_library(21, 1500, 10, (150, 170), max_err=3, n_rate=0.02, rc_rate=0.4, shared_flank=40)
    _compare(engine, reads, names, 0.99)
    reads, names = _noisy_library(22, 1500, 6, (100, 110), max_err=2, indel=False, n_rate=0.03, rc_rate=0.5, shared_flank=30)
    _compare(engine, reads, names, 0.985)


def test_cluster_rejection_certificate(engine, monkeypatch):
    """Most candidate alignments are proven rejections without the dynamic program (k_cl_precheck); switching the
    certificate off must not change a single outcome -- both runs are compared with the oracle -- and it must actually fire."""
    reads, names = _noisy_library(31, 2500, 40, (250, 300), max_err=5, n_rate=0.004)
    _, st = _compare(engine, reads, names, 0.99)
    assert st["cl_certified"] > 5 * st["cl_alignments"] > 0
    monkeypatch.setenv("ITSX_CL_NOPRECHECK", "1")
    _, st0 = _compare(engine, reads, names, 0.99)
    assert st0["cl_certified"] == 0 and st0["cl_alignments"] > 3 * st["cl_alignments"]
    # the score pass (the optimal score as the certificate's bound, for candidates whose best diagonal says nothing) on its own switch
    monkeypatch.delenv("ITSX_CL_NOPRECHECK")
    monkeypatch.setenv("ITSX_CL_NOSCORE", "1")
    _, st1 = _compare(engine, reads, names, 0.99)
    assert st["cl_alignments"] < st1["cl_alignments"] < st0["cl_alignments"]


def test_cluster_score_pass_pairs_of_unlike_targets(engine, monkeypatch):
    """The score pass runs two candidates of a strand at a time, their scores in the 16-bit halves of one register: targets of very
    different lengths in one pair, IUPAC codes on both sides, queries at the last length one pass of 8 rows per lane holds (511) and
    the first that needs 10 (512); many small unrelated families, so that most candidates are 'too weak' for the diagonal bound."""
    rng = np.random.default_rng(77)
    reads, names = [], []
    for lo, hi, seed in ((120, 180, 1), (300, 330, 2), (505, 512, 3)):
        r, n = _noisy_library(90 + seed, 700, 90, (lo, hi), max_err=3, n_rate=0.003, rc_rate=0.4, shared_flank=20)
        reads += r
        names += ["%s;%d" % (x, seed) for x in n]
    iupac = "RYSWKMBDHVN"
    for i in range(0, len(reads), 7):                            # degenerate codes in every seventh read
        s = list(reads[i])
        for _ in range(3):
            s[int(rng.integers(0, len(s)))] = iupac[int(rng.integers(0, len(iupac)))]
        reads[i] = "".join(s)
    perm = rng.permutation(len(reads))
    reads = [reads[i] for i in perm]
    names = [names[i] for i in perm]
    assert max(map(len, reads)) >= 512 and any(len(r) == 511 for r in reads)
    _, st = _compare(engine, reads, names, 0.99)
    monkeypatch.setenv("ITSX_CL_NOSCORE", "1")
    _, st1 = _compare(engine, reads, names, 0.99)
    assert st1["cl_alignments"] > 2 * st["cl_alignments"] > 0   # the pass did take most of them
    monkeypatch.delenv("ITSX_CL_NOSCORE")
    monkeypatch.setenv("ITSX_CL_ROWS", "8")                      # 8 rows per lane although reads of 512+ exist: those queries skip the pass
    _compare(engine, reads, names, 0.99)


def test_cluster_plus_strand_only_and_no_names(engine):
    reads, names = _noisy_library(13, 1200, 30, (200, 240))
    _compare(engine, reads, names, 0.99, strand_both=False)
    _compare(engine, reads, None, 0.99)


def test_cluster_multipass_alignment(engine, monkeypatch):
    # 5 rows per lane cover 320 DP rows: 400-base reads need two passes over the boundary-row scratch
    monkeypatch.setenv("ITSX_CL_ROWS", "5")
    reads, names = _noisy_library(14, 500, 12, (380, 420))
    _compare(engine, reads, names, 0.985)
    monkeypatch.delenv("ITSX_CL_ROWS")
    reads, names = _noisy_library(15, 300, 8, (650, 700))
    _compare(engine, reads, names, 0.99)


@pytest.mark.parametrize("heavy", ["0", "3"])
def test_cluster_conserved_words_by_bitmap_or_by_list(engine, heavy, monkeypatch):
    # words held by >= 128 strands of a window are counted through strand bitmaps (bit-sliced adders) instead of their lists;
    # with no bitmaps at all (0) and with only three of them (the rest of the conserved words keep their lists) the outcome is the same
    monkeypatch.setenv("ITSX_CL_HEAVY", heavy)
    reads, names = _noisy_library(51, 2500, 6, (180, 230), max_err=3, n_rate=0.003, rc_rate=0.2, shared_flank=50)
    _compare(engine, reads, names, 0.985)


def test_cluster_long_reads_take_their_words_in_batches(engine):
    # a centroid's words are taken 1 024 at a time (LDS); 1 300-base reads have ~1 290, most of them conserved inside their
    # (large) family, so both batches go through bitmaps AND lists
    reads, names = _noisy_library(61, 420, 2, (1280, 1320), max_err=4, indel=True, n_rate=0.001, rc_rate=0.2, shared_flank=80)
    o, st = _compare(engine, reads, names, 0.99)
    assert o["n_centroids"] < 200


def test_cluster_index_growth(engine, monkeypatch):
    # more centroid words than the initial capacity of the word pool: the pool is doubled (with a copy) on the way
    monkeypatch.setenv("ITSX_CL_CAPACITY", "2048")
    rng = np.random.default_rng(5)
    acgt = np.array(list("ACGT"))
    reads = ["".join(acgt[rng.integers(0, 4, 60)]) for _ in range(7000)]          # unrelated: (almost) every read a centroid
    reads += reads[:500]                                                          # and some duplicates that must find theirs
    names = ["g%05d" % i for i in range(len(reads))]
    o, st = _compare(engine, reads, names, 0.97)
    assert o["n_centroids"] >= 6900


def test_cluster_edge_cases(engine):
    # short reads vanish; a read of only N has no words and becomes its own centroid; duplicates join at 100 %
    base = "ACGTTGCAAGCTTAGGCTAACGGTCAGTCCATGGATCAGGCTTAAGCCGGTATCGATTACGGCAT" * 3
    reads = [base, base, base[:20], "N" * 40, base[::-1].translate(_RC), base[:100] + "A" + base[101:], "N" * 40]
    names = ["r%d" % i for i in range(len(reads))]
    o, _ = _compare(engine, reads, names, 0.99)
    assert o["rep_of"].tolist() == [0, 0, -1, 3, 0, 0, 6] and o["strand"][4] == -1
    assert o["pct_id"][1] == 100.0 and o["pct_id"][5] == 100.0 * (len(base) - 1) / len(base)
    # empty input and nothing kept
    engine.set_reads([], [])
    assert engine.cluster(0.99) == 0
    engine.set_reads(["ACGT"], ["a"])
    assert engine.cluster(0.99) == 0


def test_cluster_then_search_and_files(engine, mini_hmm_text, tmp_path):
    """SeqSample.cluster -> uc.txt / rep.fa -> Dedup.parse semantics, then the HMM stages run on the centroids."""
    blob, offs = synth.make_reads(mini_hmm_text, 1500, seed=5, sub_rate=0.004)
    reads = synth.to_strings(blob, offs)
    names = ["q%05d" % ((i * 7919) % 100000) for i in range(len(reads))]
    engine.load_profiles(text=mini_hmm_text)
    engine.set_reads(reads, names)
    nuniq = engine.derep()
    ncl = engine.cluster(0.99)
    assert 0 < ncl < nuniq
    codes, off = orc.digitize(reads)
    o = orc.cluster(codes, off, names, 0.99)
    uc = str(tmp_path / "uc.txt")
    rep = str(tmp_path / "rep.fa")
    engine.write_uc(uc)
    engine.write_rep_fasta(rep)
    match = {}
    rows = [ln.rstrip("\n").split("\t") for ln in open(uc)]
    for ll in rows:                                   # Dedup.parse (itsxpress/SeqSample.py:542-562)
        if ll[0] == "S":
            match[ll[8]] = ll[8]
        elif ll[0] == "H":
            match[ll[8]] = ll[9]
    assert match == {names[i]: names[int(o["rep_of"][i])] for i in range(len(reads)) if o["rep_of"][i] >= 0}
    assert [ll[8] for ll in rows if ll[0] in "SH"] == [names[i] for i in o["order"]]
    assert sum(ll[0] == "C" for ll in rows) == ncl
    heads = [ln[1:].strip() for ln in open(rep) if ln[0] == ">"]
    assert heads == [names[i] for i in o["order"] if o["rep_of"][i] == i]
    # the centroids are what gets searched, and every read inherits its centroid's coordinates
    engine.search()
    engine.finalize()
    start, stop, tlen, ind = engine.trim_coords("3_", "4_")
    rs, re_, rt, _ = engine.rep_coords("3_", "4_")
    _, _, uniq_of = engine.get_derep()
    assert np.array_equal(start, rs[uniq_of]) and np.array_equal(stop, re_[uniq_of]) and (start >= 0).sum() > 1000


def _low_complexity_reads(seed, n):
    """homopolymer / microsatellite stretches inside amplicon-like reads, plus the edge shapes of the windowed filter"""
    rng = np.random.default_rng(seed)
    rnd = lambda k: "".join(rng.choice(list("ACGT"), int(k)))
    tmpl = []
    for t in range(8):
        core = [rnd(40), "A" * int(rng.integers(10, 40)), rnd(35), "CA" * int(rng.integers(6, 30)), rnd(30), "TTG" * int(rng.integers(5, 14)), rnd(45),
                "ACGTTGCA" * int(rng.integers(0, 6)), rnd(rng.integers(0, 70))]
        tmpl.append("".join(core[k] for k in rng.permutation(len(core))))
    reads, names = [], []
    for i in range(n):
        s = list(tmpl[int(rng.integers(0, len(tmpl)))])
        for _ in range(int(rng.integers(0, 5))):
            s[int(rng.integers(0, len(s)))] = str(rng.choice(list("ACGTN")))
        s = "".join(s)
        if rng.random() < 0.3:
            s = s[::-1].translate(_RC)
        reads.append(s)
        names.append("d%05d" % int(rng.integers(0, 50000)))
    return reads, names


def test_dust_masks_equal_the_oracle(engine):
    """k_dust (one wave per read, one lane per window start) == orc_dust on every base: window advance, the pull-forward after a
    masked first half, reads shorter than a window, N runs (N counts as A)"""
    rng = np.random.default_rng(3)
    rnd = lambda k: "".join(rng.choice(list("ACGT"), int(k)))
    reads, _ = _low_complexity_reads(5, 300)
    reads += [rnd(200), "T" * 30, "A" * 8, rnd(5), rnd(7), rnd(8), rnd(63) + "G" * 20, rnd(64), rnd(65), rnd(33) + "N" * 30 + rnd(40), "AC" * 400,
              rnd(31) + "GA" * 9 + rnd(28) + "C" * 11 + rnd(70), rnd(95) + "T" * 33, "G" * 33 + rnd(95), rnd(1500) + "CAG" * 30 + rnd(700)]
    engine.set_reads(reads)
    got = engine.debug_dust([len(r) for r in reads])
    n_masked = 0
    for r, g in zip(reads, got):
        exp = orc.dust(r)
        assert np.array_equal(g, exp), r[:80]
        n_masked += int(exp.sum())
    assert n_masked > 10000


def test_cluster_with_dust_masked_seeds(engine, monkeypatch):
    """vsearch's default --qmask dust / --dbmask dust (SeqSample.py:147-161 passes neither option): engine == oracle with the
    masking on (the default) and off (ITSX_QMASK=none / ORC_QMASK=none), and it changes outcomes on low-complexity reads"""
    reads, names = _low_complexity_reads(6, 1500)
    o1, _ = _compare(engine, reads, names, 0.97)
    monkeypatch.setenv("ITSX_QMASK", "none")
    monkeypatch.setenv("ORC_QMASK", "none")
    o0, _ = _compare(engine, reads, names, 0.97)
    assert o1["n_alignments"] != o0["n_alignments"]
