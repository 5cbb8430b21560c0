"""Two-sided sharing (round 6; csrc/k_share.hip: k_join_*, csrc/k_lazy.hip: k_bwd_bound and the join in k_fwd_bound): a representative
whose last blocks an earlier representative of its length ends with too stops its Forward chain there and takes the rest of the sum
over paths from the Backward state that representative's chain saved.  hmmsearch scores every target from its first to its last
residue (itsxpress/SeqSample.py:191-209); pass A's score is only a BOUND for the selection, so the schedule may change its rounding
but nothing else: every joined score within 2e-3 nats of the unshared kernel's (ITSX_SHARE_CHECK=1), every other score bitwise, the
bound within 1e-3 nats of HMMER's own Forward arithmetic (ITSX_LAZY_CHECK_BOUND=1), and every coordinate the full table's and the
one-sided schedule's.  `pytest -m gpu`."""
import numpy as np
import pytest

import synth
from test_gpu_compact import _same
from test_gpu_parity import _its2_subset
from test_gpu_share import _bench_reads, _edge_reads, _run

pytestmark = pytest.mark.gpu


def test_two_sided_scores_and_coordinates(engine, t_hmm_text, monkeypatch):
    """configs[2]'s shape: most representatives join, the rows walked fall below the one-sided tree's, no score leaves its tolerance,
    coordinates equal to the unshared search, the one-sided search and the full table"""
    hmm = _its2_subset(t_hmm_text, 40, 40)
    seqs = _bench_reads(t_hmm_text, 30000)
    ref, st0 = _run(engine, hmm, seqs, "lazy", {"ITSX_SHARE": 0}, monkeypatch)
    one, st1 = _run(engine, hmm, seqs, "lazy", {"ITSX_SHARE_TWO": 0, "ITSX_SHARE_CHECK": 1}, monkeypatch)
    assert st1["share_B"] == 32 and not st1["two_sided"] and st1["share_mismatch"] == 0 and st1["bwd_rows"] == 0
    got, st = _run(engine, hmm, seqs, "lazy", {"ITSX_SHARE_CHECK": 1, "ITSX_LAZY_CHECK_BOUND": 1}, monkeypatch)
    assert st["two_sided"] == 1 and st["n_joined"] > 0.5 * st["n_unique"] and st["bwd_chains"] > 0 and st["gamma_nodes"] > 0
    assert st["share_mismatch"] == 0 and 0 < st["join_maxdiff"] < 2e-3
    assert st["lazy_bound_maxdiff"] < 1e-3
    # rows: per representative and in lane-rows (Forward + Backward chains) well below the one-sided tree's
    assert st["two_fwd_rows"] + st["two_bwd_rows"] < 0.75 * (st1["msv_rows"] // st1["n_profiles"])
    assert st["bound_rows"] + st["bwd_rows"] < 0.75 * st1["bound_rows"]
    assert st["n_past_msv"] == st0["n_past_msv"]
    assert _same(ref, got) and _same(one, got)
    full, _ = _run(engine, hmm, seqs, "full", {}, monkeypatch)
    assert _same(full, got)


@pytest.mark.parametrize("env", [{"ITSX_SHARE_B": 16}, {"ITSX_SHARE_B": 64}, {"ITSX_SHARE_B": 128},
                                 {"ITSX_SHARE_GB": 0.02},                       # many batches
                                 {"ITSX_SHARE_GB": 0.0005},                     # ... and the profiles in ranges
                                 {"ITSX_CHUNK_UNIQUES": 1777},                  # nothing is shared across chunks
                                 {"ITSX_CHUNK_UNIQUES": 1777, "ITSX_SHARE_GB": 0.004},
                                 {"ITSX_SHARE_GB": 0.02, "ITSX_SHARE_FWD_STREAMS": 2},
                                 {"ITSX_SHARE_GB": 0.02, "ITSX_SHARE_FWD_STREAMS": 1}])
def test_two_sided_batches_splits_chunks_and_block_sizes(engine, t_hmm_text, monkeypatch, env):
    hmm = _its2_subset(t_hmm_text, 12, 12)
    seqs = _bench_reads(t_hmm_text, 9000, seed=6)
    ref, _ = _run(engine, hmm, seqs, "lazy", {"ITSX_SHARE": 0, **{k: v for k, v in env.items() if k == "ITSX_CHUNK_UNIQUES"}}, monkeypatch)
    got, st = _run(engine, hmm, seqs, "lazy", {"ITSX_SHARE_CHECK": 1, "ITSX_LAZY_CHECK_BOUND": 1, "ITSX_SHARE_MIN": 0, **env}, monkeypatch)
    assert st["two_sided"] == 1 and st["n_joined"] > 0 and st["share_mismatch"] == 0 and st["join_maxdiff"] < 2e-3 and st["lazy_bound_maxdiff"] < 1e-3
    assert _same(ref, got)


def test_two_sided_edges(engine, t_hmm_text, monkeypatch):
    """families that differ in the first / last base and at block boundaries, N's before / at / after a branch point on either side,
    lengths that are multiples of the block, reads shorter than a block: joins at every level incl. chains without a row of their own"""
    hmm = _its2_subset(t_hmm_text, 6, 6)
    lm = synth.consensus_motifs(t_hmm_text, "3_")[0]
    rm = synth.consensus_motifs(t_hmm_text, "4_")[0]
    seqs = _edge_reads(np.random.default_rng(77), lm, rm)
    # recombinants: the first half of one family member with the second half of another (prefix of p, suffix of q: no row of their own)
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for L in (300, 448):
        base = acgt[rng.integers(0, 4, L)].copy()
        base[30:75] = np.frombuffer(lm.encode(), np.uint8); base[L - 60:L - 15] = np.frombuffer(rm.encode(), np.uint8)
        a = base.copy(); a[L - 5] = acgt[(np.searchsorted(acgt, a[L - 5]) + 1) % 4]
        b = base.copy(); b[3] = acgt[(np.searchsorted(acgt, b[3]) + 1) % 4]
        c = b.copy(); c[L - 5] = a[L - 5]
        seqs += [bytes(x).decode() for x in (base, a, b, c)]
    for B in (16, 32, 64):
        ref, _ = _run(engine, hmm, seqs, "lazy", {"ITSX_SHARE": 0}, monkeypatch)
        got, st = _run(engine, hmm, seqs, "lazy", {"ITSX_SHARE_CHECK": 1, "ITSX_LAZY_CHECK_BOUND": 1, "ITSX_SHARE_MIN": 0, "ITSX_SHARE_B": B}, monkeypatch)
        assert st["two_sided"] == 1 and st["n_joined"] > 20 and st["share_mismatch"] == 0 and st["join_maxdiff"] < 2e-3 and st["lazy_bound_maxdiff"] < 1e-3, (B, st)
        assert _same(ref, got), B


def test_two_sided_on_ccs_length_reads(engine, t_hmm_text, monkeypatch):
    """2.3-5 kb families: both trees hold 63 blocks -- the Forward chains of long reads join in their last 2 016 rows only -- and the
    joined bound must still bound HMMER's Forward (lazy == full)"""
    rng = np.random.default_rng(78)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    lm = synth.consensus_motifs(t_hmm_text, "3_")[0]
    rm = synth.consensus_motifs(t_hmm_text, "4_")[0]
    seqs = []
    for L in (2300, 3100, 5000):
        base = acgt[rng.integers(0, 4, L)].copy()
        for a in range(200, L - 400, 700):
            base[a:a + 45] = np.frombuffer(lm.encode(), np.uint8)
            base[a + 250:a + 295] = np.frombuffer(rm.encode(), np.uint8)
        seqs.append(bytes(base).decode())
        for j in range(40):
            v = base.copy()
            for pos in rng.integers(0, L, 1 + j % 3):
                v[pos] = acgt[(np.searchsorted(acgt, v[pos]) + 1) % 4]
            seqs.append(bytes(v).decode())
    hmm = _its2_subset(t_hmm_text, 8, 8)
    full, _ = _run(engine, hmm, seqs, "full", {"ITSX_SHARE": 0}, monkeypatch)
    got, st = _run(engine, hmm, seqs, "lazy", {"ITSX_SHARE_CHECK": 1, "ITSX_LAZY_CHECK_BOUND": 1, "ITSX_SHARE_MIN": 0}, monkeypatch)
    assert st["two_sided"] == 1 and st["n_joined"] > 30 and st["share_mismatch"] == 0 and st["join_maxdiff"] < 2e-3
    assert st["lazy_bound_maxdiff"] < 0.5 * 0.02 * 0.6931           # half the margin, as for the one-sided chains (test_gpu_lazy.py)
    assert _same(full, got)
