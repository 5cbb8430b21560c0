"""Prefix sharing (csrc/k_share.hip): the MSV filter and the lazy stage's score-only Forward pass start a representative's rows from
the state another representative of the same length saved where their sequences part.  hmmsearch scores every target from its first
residue (itsxpress/SeqSample.py:191-209), so the shared schedule must change NOTHING: every MSV cell and every Forward score bitwise
those of the unshared kernels (ITSX_SHARE_CHECK=1 runs both and counts the differences), every coordinate the unshared search's and
the CPU oracle's -- on one batch and many, with the profiles split, across chunks, at every block size, with N's, with reads shorter
than a block, with reads past the tree's 63 blocks.  `pytest -m gpu`."""
import os

import numpy as np
import pytest

import orc
import synth
from test_gpu_compact import PAIRS, _same
from test_gpu_parity import _its2_subset

pytestmark = pytest.mark.gpu


def _run(engine, hmm, seqs, mode, env, monkeypatch, names=None):
    for k in ("ITSX_SHARE", "ITSX_SHARE_B", "ITSX_SHARE_GB", "ITSX_SHARE_MIN", "ITSX_SHARE_CHECK", "ITSX_CHUNK_UNIQUES", "ITSX_SHARE_FWD_STREAMS", "ITSX_SHARE_TWO"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    engine.set_rows_mode(mode)
    try:
        engine.load_profiles(text=hmm)
        engine.set_reads(seqs, names)
        engine.derep()
        engine.search()
        engine.finalize()
        st = engine.stats()
        return [tuple(a.copy() for a in engine.trim_coords(l, r)) for l, r in PAIRS], st
    finally:
        engine.set_rows_mode(None)


def _bench_reads(t_hmm_text, n, seed=5, **kw):
    blob, offs = synth.make_reads(t_hmm_text, n, config=3, seed=synth.SEED + seed, fixed_len=0, len_range=(300, 580), **kw)
    return synth.to_strings(blob, offs)


@pytest.mark.parametrize("mode", ["lazy", "compact"])
def test_shared_schedule_is_bitwise_the_unshared_one(engine, t_hmm_text, monkeypatch, mode):
    """configs[2]'s shape: a third of the rows shared, not one MSV cell or Forward score differs, coordinates equal"""
    hmm = _its2_subset(t_hmm_text, 40, 40)
    seqs = _bench_reads(t_hmm_text, 20000)
    ref, st0 = _run(engine, hmm, seqs, mode, {"ITSX_SHARE": 0}, monkeypatch)
    assert st0["share_B"] == 0 and st0["msv_rows"] == st0["msv_rows_full"]
    got, st = _run(engine, hmm, seqs, mode, {"ITSX_SHARE_CHECK": 1}, monkeypatch)
    assert st["share_B"] == 32 and st["share_chains"] > 0 and st["share_nodes"] > 0
    assert st["share_mismatch"] == 0
    assert st["share_frac"] > 0.2 and st["msv_rows"] < 0.8 * st["msv_rows_full"]
    assert st["n_past_msv"] == st0["n_past_msv"] and st["n_pairs"] == st0["n_pairs"]
    if mode == "lazy":
        assert 0 < st["bound_rows"] < 0.85 * st["bound_rows_full"]
        # (a joined pair's bound is the same sum over paths in another order of operations: a tenth may round the other way for a few)
        assert abs(st["n_lazy_evaluated"] - st0["n_lazy_evaluated"]) <= (0 if not st["two_sided"] else 1e-3 * st0["n_lazy_evaluated"] + 2)
    assert _same(ref, got)


@pytest.mark.parametrize("env", [{"ITSX_SHARE_B": 16}, {"ITSX_SHARE_B": 64}, {"ITSX_SHARE_B": 128},
                                 {"ITSX_SHARE_GB": 0.02},                       # many batches
                                 {"ITSX_SHARE_GB": 0.0005},                     # ... and the profiles in ranges
                                 {"ITSX_CHUNK_UNIQUES": 1777},                  # nothing is shared across chunks
                                 {"ITSX_CHUNK_UNIQUES": 1777, "ITSX_SHARE_GB": 0.004},
                                 {"ITSX_SHARE_GB": 0.02, "ITSX_SHARE_FWD_STREAMS": 2},      # the Forward pass's batches on two streams (the default
                                 {"ITSX_SHARE_GB": 0.02, "ITSX_SHARE_FWD_STREAMS": 1}])     # under 2 M representatives without a slot budget) / on one
def test_batches_splits_chunks_and_block_sizes(engine, t_hmm_text, monkeypatch, env):
    hmm = _its2_subset(t_hmm_text, 12, 12)
    seqs = _bench_reads(t_hmm_text, 9000, seed=6)
    ref, _ = _run(engine, hmm, seqs, "lazy", {"ITSX_SHARE": 0, **{k: v for k, v in env.items() if k == "ITSX_CHUNK_UNIQUES"}}, monkeypatch)
    got, st = _run(engine, hmm, seqs, "lazy", {"ITSX_SHARE_CHECK": 1, "ITSX_SHARE_MIN": 0, **env}, monkeypatch)
    assert st["share_B"] == env.get("ITSX_SHARE_B", 32) and st["share_mismatch"] == 0 and st["share_chains"] > 0
    if "ITSX_SHARE_GB" in env:
        assert st["share_batches"] > 1
    assert _same(ref, got)


def _edge_reads(rng, motif_l, motif_r):
    """families that exercise the tree's edges: members that differ in the first / last base, at block boundaries, by an N before /
    at / after a branch point; lengths that are multiples of the block; reads shorter than a block; equal prefixes at other lengths"""
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for L in (17, 31, 32, 33, 64, 65, 96, 128, 255, 256, 257, 300, 448):
        base = acgt[rng.integers(0, 4, L)].copy()
        if L >= 140:
            base[30:75] = np.frombuffer(motif_l.encode(), np.uint8)
            base[L - 60:L - 15] = np.frombuffer(motif_r.encode(), np.uint8)
        fam = [base]
        for pos in sorted({0, 1, 15, 16, 31, 32, 33, 63, 64, 65, L // 2, L - 2, L - 1} & set(range(L))):
            v = base.copy(); v[pos] = acgt[(np.searchsorted(acgt, v[pos]) + 1) % 4]; fam.append(v)
            w = v.copy(); w[min(L - 1, pos + 40)] = ord("N"); fam.append(w)       # an N below the branch point
            x = base.copy(); x[pos] = ord("N"); fam.append(x)                     # an N where others branch
        for k in (1, 2, 40):                                                       # the same prefix at another length
            if L - k > 0:
                fam.append(base[:L - k])
        fam.append(np.concatenate([base, base[:7]]))
        out += fam
    seqs = [bytes(s).decode() for s in out]
    order = rng.permutation(len(seqs))
    return [seqs[i] for i in order]


def test_edges_against_the_oracle(engine, t_hmm_text, monkeypatch):
    hmm = _its2_subset(t_hmm_text, 6, 6)
    lm = synth.consensus_motifs(t_hmm_text, "3_")[0]
    rm = synth.consensus_motifs(t_hmm_text, "4_")[0]
    seqs = _edge_reads(np.random.default_rng(77), lm, rm)
    codes, o = orc.digitize(seqs)
    nc, orep, ostrand = orc.derep(codes, o)
    seeds = [i for i in range(len(seqs)) if orep[i] == i]
    c2, o2 = orc.digitize([seqs[i] for i in seeds])
    res = orc.SearchResult(orc.HmmSet(text=hmm), c2, o2, threads=os.cpu_count() or 4, keep_trace=0)
    uniq = np.cumsum(np.asarray(orep) == np.arange(len(seqs))) - 1
    uo = uniq[np.maximum(orep, 0)]
    us, ue, ut, ui = res.positions("3_", "4_")
    exp = np.stack([us[uo], ue[uo], ut[uo], ui[uo]])
    for B in (16, 32, 64):
        for mode in ("lazy", "full"):
            got, st = _run(engine, hmm, seqs, mode, {"ITSX_SHARE_CHECK": 1, "ITSX_SHARE_MIN": 0, "ITSX_SHARE_B": B}, monkeypatch)
            assert st["share_B"] == B and st["share_mismatch"] == 0 and st["share_chains"] > 50
            assert np.array_equal(exp, np.stack(got[0])), (B, mode)


def test_reads_past_the_trees_depth(engine, t_hmm_text, monkeypatch):
    """CCS-length families (2.3-5 kb): the tree holds 63 blocks, the rows past it are every chain's own; and the bound kernel on rows
    that continue from a saved state must still bound HMMER's Forward (lazy == full)"""
    rng = np.random.default_rng(78)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    lm = synth.consensus_motifs(t_hmm_text, "3_")[0]
    rm = synth.consensus_motifs(t_hmm_text, "4_")[0]
    seqs = []
    for L in (2300, 3100, 5000):
        base = acgt[rng.integers(0, 4, L)].copy()
        for a in range(200, L - 400, 700):                                         # several partial / full copies: multidomain targets
            base[a:a + 45] = np.frombuffer(lm.encode(), np.uint8)
            base[a + 250:a + 295] = np.frombuffer(rm.encode(), np.uint8)
        for j in range(40):
            v = base.copy()
            for pos in rng.integers(0, L, 3):
                v[pos] = acgt[(np.searchsorted(acgt, v[pos]) + 1 + rng.integers(0, 3)) % 4]
            seqs.append(bytes(v).decode())
    hmm = _its2_subset(t_hmm_text, 8, 8)
    ref, _ = _run(engine, hmm, seqs, "full", {"ITSX_SHARE": 0}, monkeypatch)
    for B in (16, 32):
        got, st = _run(engine, hmm, seqs, "lazy", {"ITSX_SHARE_CHECK": 1, "ITSX_SHARE_MIN": 0, "ITSX_SHARE_B": B}, monkeypatch)
        assert st["share_B"] == B and st["share_mismatch"] == 0 and st["share_chains"] > 60
        assert _same(ref, got)


def test_random_reads_fall_back_to_the_plain_schedule(engine, mini_hmm_text, monkeypatch):
    rng = np.random.default_rng(79)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seqs = [bytes(acgt[rng.integers(0, 4, int(L))]).decode() for L in rng.integers(150, 400, 3000)]
    _, st = _run(engine, mini_hmm_text, seqs, "lazy", {}, monkeypatch)
    assert st["share_B"] == 0 and st["share_frac"] < 0.1 and st["msv_rows"] == st["msv_rows_full"]


def test_sample_batches_share_across_samples(engine, t_hmm_text, monkeypatch):
    """equal sequences of different samples are separate representatives with one prefix tree: every sample's coordinates are those
    of the unshared search"""
    hmm = _its2_subset(t_hmm_text, 10, 10)
    seqs = _bench_reads(t_hmm_text, 4000, seed=8)
    smp = (np.arange(len(seqs)) % 5).astype(np.int32)
    outs = []
    for env in ({"ITSX_SHARE": 0}, {"ITSX_SHARE_CHECK": 1, "ITSX_SHARE_MIN": 0}):
        for k in ("ITSX_SHARE", "ITSX_SHARE_CHECK", "ITSX_SHARE_MIN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, str(v))
        engine.set_rows_mode("lazy")
        try:
            engine.load_profiles(text=hmm)
            engine.set_reads(seqs)
            engine.set_samples(smp, 5)
            engine.derep()
            engine.search()
            engine.finalize()
            st = engine.stats()
            outs.append(([a.copy() for a in engine.trim_coords("3_", "4_")], st))
        finally:
            engine.set_rows_mode(None)
    assert outs[1][1]["share_B"] == 32 and outs[1][1]["share_mismatch"] == 0
    assert all(np.array_equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))
