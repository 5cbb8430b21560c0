"""The lazy domain stage (itsx_set_rows_mode(ctx, ITSX_ROWS_LAZY); csrc/k_lazy.hip): pairs that cannot win ItsPosition's argmax
(itsxpress/SeqSample.py:400-461) are not evaluated past their Forward score.  The per-read coordinates and the "sequence has a
row" flag must be EXACTLY those of the full table -- on every prefix pair the reference uses, on one chunk and many, on damaged
reads whose rows sit near the thresholds, against the CPU oracle, after a domZ exchange, for sample batches -- and the stage must
really skip work.  `pytest -m gpu`."""
import os

import numpy as np
import pytest

import orc
import synth
from test_gpu_compact import PAIRS, _same, _weak
from test_gpu_parity import _its2_subset

pytestmark = pytest.mark.gpu


def _coords(engine, hmm, seqs, mode, domE=10.0, samples=None):
    engine.set_rows_mode(mode)
    try:
        engine.load_profiles(text=hmm)
        engine.set_reads(seqs)
        if samples is not None:
            engine.set_samples(samples[0], samples[1])
        engine.derep()
        engine.search()
        engine.finalize(domE=domE)
        st = engine.stats()
        return [tuple(a.copy() for a in engine.trim_coords(l, r)) for l, r in PAIRS], st
    finally:
        engine.set_rows_mode(None)


@pytest.mark.parametrize("chunk", [None, "53"])
def test_lazy_coordinates_equal_the_full_table(engine, mini_hmm_text, t_hmm_text, monkeypatch, chunk):
    rng = np.random.default_rng(23)
    blob, offs = synth.make_reads(mini_hmm_text, 1500, seed=91, fixed_len=0, len_range=(200, 520))
    seqs = _weak(synth.to_strings(blob, offs), rng)
    hmm = mini_hmm_text + _its2_subset(t_hmm_text, 25, 25)            # 1_ 2_ 3_ 4_ profiles, families of near-identical ones
    if chunk:
        monkeypatch.setenv("ITSX_CHUNK_UNIQUES", chunk)
    monkeypatch.delenv("ITSX_COMPACT_ROWS", raising=False)
    ref, st0 = _coords(engine, hmm, seqs, "full")
    assert sum(int(((c[0] >= 0) | (c[1] >= 0)).sum()) for c in ref) > 1500
    got, st1 = _coords(engine, hmm, seqs, "lazy")
    assert st1["lazy"] == 1 and st1["n_lazy_reruns"] == 0
    assert _same(ref, got)
    # work really skipped: far fewer pairs through Backward than past the MSV filter
    assert 0 < st1["n_lazy_evaluated"] < 0.5 * st1["n_past_msv"]
    assert st1["n_lazy_round1"] <= st1["n_lazy_evaluated"]
    # another domE inside the compaction's assumptions
    ref2, _ = _coords(engine, hmm, seqs, "full", domE=0.05)
    got2, _ = _coords(engine, hmm, seqs, "lazy", domE=0.05)
    assert _same(ref2, got2)


def test_lazy_equals_the_oracle_on_the_bench_shape(engine, t_hmm_text):
    """configs[2]'s shape (merged reads of 300-580 bases, the stand-in taxon's 155 ITS2 profiles): lazy coordinates == the CPU oracle's"""
    hmm = _its2_subset(t_hmm_text, 10 ** 6, 10 ** 6)
    blob, offs = synth.make_reads(t_hmm_text, 1200, config=3, seed=synth.SEED + 3, fixed_len=0, len_range=(300, 580))
    seqs = synth.to_strings(blob, offs)
    got, st = _coords(engine, hmm, seqs, "lazy")
    assert st["n_lazy_evaluated"] < 0.15 * st["n_past_msv"]           # scripts/lazy_bound.py: 3.7 % on this workload
    codes, o = orc.digitize(seqs)
    nc, orep, ostrand = orc.derep(codes, o)
    seeds = [i for i in range(len(seqs)) if orep[i] == i]
    c2, o2 = orc.digitize([seqs[i] for i in seeds])
    res = orc.SearchResult(orc.HmmSet(text=hmm), c2, o2, threads=os.cpu_count() or 4, keep_trace=0)
    uniq = np.cumsum(np.asarray(orep) == np.arange(len(seqs))) - 1
    uo = uniq[np.maximum(orep, 0)]
    us, ue, ut, ui = res.positions("3_", "4_")
    exp = np.stack([us[uo], ue[uo], ut[uo], ui[uo]])
    assert np.array_equal(exp, np.stack(got[0]))


def test_lazy_follows_a_domz_exchange(engine, mini_hmm_text):
    """multi-GPU: the counters are summed over the ranks between search and finalize.  After a lazy search they are bounds (lower
    half, upper half); a 50 000-fold inflation of both un-reports weak rows, and the lazy result must still equal the full table's
    -- or say that rows are pending (then the driver repeats the search in compact mode)."""
    rng = np.random.default_rng(29)
    blob, offs = synth.make_reads(mini_hmm_text, 800, seed=72, fixed_len=0, len_range=(200, 420))
    seqs = _weak(synth.to_strings(blob, offs), rng, frac=0.5)
    out = []
    for mode in ("full", "lazy"):
        engine.set_rows_mode(mode)
        engine.load_profiles(text=mini_hmm_text)
        engine.set_reads(seqs)
        engine.derep()
        engine.search()
        z = engine.get_domz()
        if mode == "full":
            assert z.size == engine.n_profiles
            exact = z.copy()
            engine.set_domz(z * 50000)
        else:
            assert z.size == 2 * engine.n_profiles
            lo, hi = z[:engine.n_profiles], z[engine.n_profiles:]
            assert (lo <= exact).all() and (exact <= hi).all()         # bounds of the exact counters
            # what an all-reduce over 50 000 identical ranks would give; tight bounds (lower = upper = exact) decide every row
            engine.set_domz(np.concatenate([exact, exact]) * 50000)
        engine.finalize()
        assert engine.lazy_pending() == 0
        out.append([tuple(a.copy() for a in engine.trim_coords(l, r)) for l, r in PAIRS])
    engine.set_rows_mode(None)
    assert _same(out[0], out[1])


def test_rows_that_depend_on_domz_trigger_the_full_search(engine, mini_hmm_text, monkeypatch):
    """an undecided row that could change a result: alone, the context repeats the search with every pair evaluated (same
    coordinates as the full table); with exchanged counters it reports the rows as pending and refuses coordinates"""
    from itsxpress_amd import EngineError
    rng = np.random.default_rng(31)
    blob, offs = synth.make_reads(mini_hmm_text, 600, seed=73, fixed_len=0, len_range=(200, 420))
    seqs = _weak(synth.to_strings(blob, offs), rng, frac=0.5)
    ref, _ = _coords(engine, mini_hmm_text, seqs, "full")
    monkeypatch.setenv("ITSX_LAZY_FORCE_PENDING", "3")
    got, st = _coords(engine, mini_hmm_text, seqs, "lazy")
    assert st["n_lazy_reruns"] == 1 and st["n_lazy_pending"] >= 3 and st["lazy"] == 0
    assert _same(ref, got)
    engine.set_rows_mode("lazy")
    engine.search()
    engine.set_domz(engine.get_domz())                       # counters exchanged: the driver decides
    engine.finalize()
    assert engine.lazy_pending() >= 3
    with pytest.raises(EngineError):
        engine.trim_coords("3_", "4_")
    monkeypatch.delenv("ITSX_LAZY_FORCE_PENDING")
    engine.set_rows_mode("compact")
    engine.search()
    engine.finalize()
    got2 = [tuple(a.copy() for a in engine.trim_coords(l, r)) for l, r in PAIRS]
    engine.set_rows_mode(None)
    assert _same(ref, got2)


def test_undecided_rows_are_settled_by_counting_their_profiles(engine, mini_hmm_text, t_hmm_text, monkeypatch):
    """With loose bounds on domZ (the hook inflates the upper one 500 000-fold, as a data set that much larger would) weak rows stay
    undecided; the profiles of those that matter are then counted exactly (itsx_lazy_complete: every pair of theirs evaluated) and
    the coordinates equal the full table's -- without the full search."""
    rng = np.random.default_rng(37)
    blob, offs = synth.make_reads(mini_hmm_text, 1500, seed=75, fixed_len=0, len_range=(200, 520))
    seqs = []
    for s in synth.to_strings(blob, offs):                   # heavy damage: many reads keep only rows near the Forward filter's threshold
        if rng.random() < 0.6:
            s = list(s)
            for q in rng.choice(len(s), int(len(s) * rng.uniform(0.3, 0.42)), replace=False):
                s[q] = str(rng.choice(list("ACGT")))
            s = "".join(s)
        seqs.append(s)
    hmm = mini_hmm_text + _its2_subset(t_hmm_text, 25, 25)
    monkeypatch.setenv("ITSX_LAZY_ZUB_SCALE", "500000")
    monkeypatch.setenv("ITSX_LAZY_TOPUP", "0")               # (this test is about the full count; the top-up round ahead of it has its own below)
    e = engine
    try:
        ref, _ = _coords(e, hmm, seqs, "full")
        got, st = _coords(e, hmm, seqs, "lazy")
        assert st["n_lazy_pending"] > 0 and st["n_lazy_reruns"] == 0 and st["lazy"] == 1
        assert 0 < st["n_lazy_completed_profiles"] <= e.n_profiles and st["n_lazy_completed"] > 0
        assert _same(ref, got)
        # the multi-rank protocol by hand: counters exchanged, so finalize only reports; flags -> complete -> exchange -> finalize
        e.set_rows_mode("lazy")
        e.search()
        e.set_domz(e.get_domz())
        e.finalize()
        assert e.lazy_pending() > 0
        flags = e.lazy_pending_profiles()
        assert flags.sum() == st["n_lazy_pending_profiles"] or flags.sum() > 0
        e.lazy_complete(flags)
        z = e.get_domz()
        P = e.n_profiles
        e.set_domz(z)                       # (one rank: the "sum" is the context's own counters)
        assert (z[:P][flags > 0] == z[P:][flags > 0]).all()      # counted, not bounded
        e.finalize()
        assert e.lazy_pending() == 0
        got2 = [tuple(a.copy() for a in e.trim_coords(l, r)) for l, r in PAIRS]
        assert _same(ref, got2)
    finally:
        e.set_rows_mode(None)


def test_top_up_round_settles_rows_from_below(engine, mini_hmm_text, t_hmm_text, monkeypatch):
    """Round 6: before a profile with undecided rows is counted in full, its best-bound unevaluated pairs go through the pipeline -- as many
    as the lower bound on its domZ needs to pass domE / P-value of those rows (itsx_search_finalize: lazy_topup).  Weak rows on a
    workload where nearly every pair past the filter is a reported target: the round runs, settles rows, what it cannot settle is still
    counted in full, and the coordinates are the full table's either way (ITSX_LAZY_TOPUP=0: straight to the full count)."""
    rng = np.random.default_rng(37)
    blob, offs = synth.make_reads(mini_hmm_text, 3000, seed=75, fixed_len=0, len_range=(200, 520))
    seqs = []
    for s in synth.to_strings(blob, offs):                   # heavy damage: many reads keep only rows near the Forward filter's threshold
        if rng.random() < 0.6:
            s = list(s)
            for q in rng.choice(len(s), int(len(s) * rng.uniform(0.3, 0.42)), replace=False):
                s[q] = str(rng.choice(list("ACGT")))
            s = "".join(s)
        seqs.append(s)
    hmm = mini_hmm_text + _its2_subset(t_hmm_text, 25, 25)
    monkeypatch.setenv("ITSX_LAZY_ZUB_SCALE", "500000")      # (upper bounds as of a data set that much larger: weak rows stay undecided at first)
    try:
        ref, _ = _coords(engine, hmm, seqs, "full")
        monkeypatch.setenv("ITSX_LAZY_TOPUP", "1")
        got, st = _coords(engine, hmm, seqs, "lazy")
        assert st["n_lazy_pending"] > 0 and st["n_lazy_reruns"] == 0 and st["lazy"] == 1
        assert st["n_lazy_topup"] > 0, st
        assert _same(ref, got)
    finally:
        engine.set_rows_mode(None)


def test_lazy_sample_batches(engine, mini_hmm_text, t_hmm_text):
    """per-sample batching (q2_itsxpress.py:273-333): groups are per representative and a representative belongs to one sample;
    domZ is bounded per (sample, profile)"""
    hmm = mini_hmm_text + _its2_subset(t_hmm_text, 12, 12)
    blob, offs = synth.make_reads(mini_hmm_text, 900, seed=77, fixed_len=0, len_range=(200, 420))
    seqs = synth.to_strings(blob, offs)
    smp = (np.arange(len(seqs)) % 3).astype(np.int32)
    ref, _ = _coords(engine, hmm, seqs, "full", samples=(smp, 3))
    got, st = _coords(engine, hmm, seqs, "lazy", samples=(smp, 3))
    assert st["lazy"] == 1
    assert _same(ref, got)


def test_lazy_with_all_taxa_profiles(engine, all_its2_hmm_text, t_hmm_text):
    """configs[3]'s profile set (814 profiles of 17 taxa): most pairs lose by tens of bits"""
    blob, offs = synth.make_reads(t_hmm_text, 400, config=4, seed=synth.SEED + 4, fixed_len=0, len_range=(300, 580))
    seqs = synth.to_strings(blob, offs)
    ref, st0 = _coords(engine, all_its2_hmm_text, seqs, "compact")
    got, st1 = _coords(engine, all_its2_hmm_text, seqs, "lazy")
    assert _same(ref, got)
    assert st1["n_lazy_evaluated"] < 0.2 * st1["n_past_msv"]


def test_bound_kernel_agrees_with_the_forward_parser(engine, t_hmm_text, mini_hmm_text, monkeypatch):
    """pass A's kernel (node-sequential, fused multiply-adds, rescaling at 1e20) computes the same sum over paths as
    p7_ForwardParser's striped arithmetic: scores within 1e-3 nats on every pair -- the bound's margin is 0.02 bits -- on ragged
    reads with N's and on profiles of other lengths (mini.hmm holds 1_ / 2_ models shorter than 45 nodes)"""
    monkeypatch.setenv("ITSX_LAZY_CHECK_BOUND", "1")
    for hmm, src, n in ((_its2_subset(t_hmm_text, 40, 40), t_hmm_text, 1500), (mini_hmm_text, mini_hmm_text, 800)):
        blob, offs = synth.make_reads(src, n, config=3, seed=synth.SEED + 11, fixed_len=0, len_range=(120, 900), n_rate=0.004)
        _, st = _coords(engine, hmm, synth.to_strings(blob, offs), "lazy")
        assert st["n_past_msv"] > 1000
        assert st["share_B"] == 32                        # (the chains of the prefix tree: rows that continue from a saved state)
        assert 0.0 <= st["lazy_bound_maxdiff"] < 1e-3, st["lazy_bound_maxdiff"]


def _ccs_reads(t_hmm_text, rng, lengths, per_family=6):
    """CCS-shaped targets (--trim-ccs inputs, itsxpress/SeqSample.py:48-91): kilobases of random sequence with full and PARTIAL copies of
    a left and a right motif in tandem, a few error variants of each"""
    acgt = np.frombuffer(b"ACGT", np.uint8)
    lm = [m for m in synth.consensus_motifs(t_hmm_text, "3_") if len(m) == 45]
    rm = [m for m in synth.consensus_motifs(t_hmm_text, "4_") if len(m) == 45]
    seqs = []
    for L in lengths:
        base = acgt[rng.integers(0, 4, L)].copy()
        a = int(rng.integers(50, 400))
        while a + 400 < L:
            l, r = lm[int(rng.integers(0, len(lm)))], rm[int(rng.integers(0, len(rm)))]
            cut = int(rng.integers(0, 25))                                 # a partial copy now and then
            base[a:a + 45 - cut] = np.frombuffer(l.encode(), np.uint8)[cut:]
            b = a + 45 + int(rng.integers(100, 260))
            base[b:b + 45] = np.frombuffer(r.encode(), np.uint8)
            a = b + 45 + int(rng.integers(200, 3000))
        for j in range(per_family):
            v = base.copy()
            for pos in rng.integers(0, L, 1 + j):
                v[pos] = acgt[(np.searchsorted(acgt, v[pos]) + 1 + rng.integers(0, 3)) % 4]
            if j == per_family - 1:
                v[int(rng.integers(0, L))] = ord("N")
            seqs.append(bytes(v).decode())
    return seqs


def test_lazy_equals_the_full_table_on_ccs_length_reads(engine, t_hmm_text, monkeypatch):
    """2-20 kb targets with tandem partial copies and one target at the engine's limit of 65 535 residues: the bound kernel stays within
    half the margin of HMMER's own Forward (the margin grows with the length: engine.hip, lazy_c), and the lazy coordinates are the full
    table's -- with the prefix tree (63 blocks deep) and without"""
    rng = np.random.default_rng(41)
    lengths = [2000, 2600, 3500, 5000, 7000, 9000, 12000, 16000, 20000]
    seqs = _ccs_reads(t_hmm_text, rng, lengths) + _ccs_reads(t_hmm_text, rng, [65535], per_family=2)
    hmm = _its2_subset(t_hmm_text, 8, 8)
    monkeypatch.setenv("ITSX_SHARE", "0")
    ref, _ = _coords(engine, hmm, seqs, "full")
    assert sum(int(((c[0] >= 0) | (c[1] >= 0)).sum()) for c in ref) > len(seqs) // 2
    monkeypatch.setenv("ITSX_LAZY_CHECK_BOUND", "1")
    margin_nats = lambda n: max(0.02, 6.0 * (n + 46) * 2.0 ** -24 / np.log(2.0)) * np.log(2.0)
    for share in ("0", "1"):
        monkeypatch.setenv("ITSX_SHARE", share)
        monkeypatch.setenv("ITSX_SHARE_MIN", "0")
        got, st = _coords(engine, hmm, seqs, "lazy")
        assert st["lazy"] == 1 and st["n_lazy_reruns"] == 0 and (st["share_B"] == 32) == (share == "1")
        assert 0.0 <= st["lazy_bound_maxdiff"] < 0.5 * margin_nats(65535), st["lazy_bound_maxdiff"]
        assert _same(ref, got)
    # the same bound on the 2-20 kb targets alone (what the 0.02-bit margin has to cover)
    _, st = _coords(engine, hmm, seqs[:-2], "lazy")
    assert 0.0 <= st["lazy_bound_maxdiff"] < 0.5 * margin_nats(20000), st["lazy_bound_maxdiff"]


def _without_match_to_delete(hmm_text, node, value="*"):
    """the profiles of `hmm_text` with t(M_node -> D_node+1) = 0 ('*'; or exp(-value)) -- the delete path past that node lives on D -> D alone"""
    out, k = [], None
    for ln in hmm_text.split("\n"):
        f = ln.split()
        if ln.startswith("HMM "):
            k = 0
        elif k is not None and len(f) >= 5 and f[0].isdigit():
            k = int(f[0])
        elif k == node and len(f) == 7 and f[6] != "*" and not ln.lstrip().startswith("1.38629  1.38629"):
            f[2] = value
            ln = "          " + "  ".join("%7s" % x for x in f)
            k = None
        if ln.startswith("//"):
            k = None
        out.append(ln)
    return "\n".join(out)


def test_bound_kernel_folded_and_plain_recurrences(engine, t_hmm_text, monkeypatch):
    """pass A keeps its delete cells divided by t(M -> D) / g and its match cells multiplied by g (constants folded into the table on
    the host: two operations per node fewer).  The folded kernel, the plain one (ITSX_BOUND_FOLD=0) and the plain one forced by a profile
    the fold cannot take (an interior M -> D of zero, or of 1e-13) all stay within 1e-3 nats of p7_ForwardParser's arithmetic, and the coordinates
    are the full table's every time"""
    blob, offs = synth.make_reads(t_hmm_text, 2500, config=3, seed=synth.SEED + 12, fixed_len=0, len_range=(150, 700), n_rate=0.003)
    seqs = synth.to_strings(blob, offs)
    hmm = _its2_subset(t_hmm_text, 20, 20)
    odd = _without_match_to_delete(hmm, 17)
    tiny = _without_match_to_delete(hmm, 17, "30.00000")       # 1e-13: the scaled delete cells would leave float's range
    assert odd != hmm and odd.count("*") > hmm.count("*") and tiny.count("30.00000") == 40
    monkeypatch.setenv("ITSX_LAZY_CHECK_BOUND", "1")
    for text in (hmm, odd, tiny):
        monkeypatch.delenv("ITSX_BOUND_FOLD", raising=False)
        ref, _ = _coords(engine, text, seqs, "full")
        outs = []
        for fold in (None, "0"):
            if fold is not None:
                monkeypatch.setenv("ITSX_BOUND_FOLD", fold)
            got, st = _coords(engine, text, seqs, "lazy")
            assert st["lazy"] == 1 and st["n_lazy_reruns"] == 0 and st["n_past_msv"] > 5000
            assert 0.0 <= st["lazy_bound_maxdiff"] < 1e-3, (fold, st["lazy_bound_maxdiff"])
            assert _same(ref, got)
            outs.append(st["lazy_bound_maxdiff"])
        if text is hmm:
            assert outs[0] != outs[1]                     # (two different kernels ran)
        else:
            assert outs[0] == outs[1]                     # (the plain one both times: the fold refused the profile set)
