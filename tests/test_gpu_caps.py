"""Nothing is refused, nothing is capped.  hmmsearch has no limit on the envelopes of a target or on the bookkeeping of a region's
traceback ensemble (p7_domaindef.c / p7_spensemble.c grow their lists; reference call site itsxpress/SeqSample.py:191-209).  The
engine's fast kernels have (8 region slots per (representative, profile); per multidomain region 8 domains in one sampled path, 512
distinct sampled tuples, 4 envelopes) -- and what does not fit them takes the overflow paths: an overflow list for the regions past a
pair's slots, a second run of the ensemble with per-region arrays sized by the region's length (k_ensemble.hip: k_mr_trace<true>).
One adversarial read per former limit, each found by a random search over tandem partial copies of a profile's consensus
(concatemer-like reads): by DEFAULT the search succeeds and every compared quantity equals the uncapped CPU oracle, bit for bit, in the
full and in the lazy rows mode.  `pytest -m gpu`."""
import numpy as np
import pytest

import orc
import synth
from test_gpu_parity import _compare

pytestmark = pytest.mark.gpu

# more than 8 domains in one sampled path (MrOut.status 2)
READ_PATH_DOMAINS = ("GCGGAACGATCTGGTCCGAGCCCGAAGCCATTAGGCCGAGGGCACGTCTGCCTGGGGAGGGCACGTCTGCCCAGGCACGTCTGCCTGGGCGTGCACGTCTGCCTGGGCGTCACCTAGGGCACGTCTG"
                     "CCTGGGCGTCACGCAGGCCGAGGGCACGTCTGCCCGAAGCCATTAGGCCGAGGCGAGGGCACGTCCGCATTAGGCCGAGGGCACGTCGGCCGCGTTTGCCTGGGCGTCACGCCGAGGGCACGTCTGCC"
                     "TGGGCGTCAGCCAGTAGGCAGAGGGCACGTCTGCCTGGGGGGCACGTCTGCCTGGGCGTCGAAAGGCTCCTACACGCCCA")
# more than 512 distinct sampled (i, j, k, m) tuples in one region (status 4)
READ_TUPLES = ("AACGCACCAAGGCGTTGTCCCCCGGGTTTAAGCACCACCCGCTGGGCCTAAGCTAGGAGGCCACCCGCTGAGTTTAAGAATCAGACGGGCGAGGCCACCCGCTGAGGCCGGATGTGTCACCCGCTGAAATTAAGC"
               "AGGATGGGCGAGGCCACCCGATGAGATTAAACAAATAAACGAGCTTACAAGGT")
# more than 4 envelopes in one region (status 7)
READ_ENVELOPES = ("AACGAACGGAGGCACGACCCCAACGCCGTTCGAGCGAGGGCAAACGCCTTTGGTCCGTCCGGCGTAGGGCACGTCTGGCTGAGTGCCGGGCGCGGATTCGCAAGTCCGCCTGGGAGTCGGCCGAGGGCTCGCC"
                  "TGCCTGGCTGCCTCTGCCTGGAAGTGACGCGGTCTATGACACCCGTACACTT")




def _coords(engine, hmm, seqs, mode):
    engine.set_rows_mode(mode)
    try:
        engine.load_profiles(text=hmm)
        engine.set_reads(seqs)
        engine.derep()
        engine.search()
        engine.finalize()
        return [tuple(a.copy() for a in engine.trim_coords(l, r)) for l, r in (("3_", "4_"), ("1_", "2_"), ("1_", "4_"))], engine.stats()
    finally:
        engine.set_rows_mode(None)


@pytest.mark.parametrize("read,what", [(READ_PATH_DOMAINS, "more than 8 domains in one sampled path"),
                                       (READ_TUPLES, "more than 512 distinct sampled tuples"),
                                       (READ_ENVELOPES, "more than 4 envelopes in one region")])
def test_ensemble_beyond_the_fast_kernels_bookkeeping_equals_the_oracle(engine, mini_hmm_text, monkeypatch, read, what):
    from test_gpu_parity import _run_both
    # ordinary companion reads: the overflow path runs beside the fast one
    blob, offs = synth.make_reads(mini_hmm_text, 20, seed=3)
    seqs = synth.to_strings(blob, offs) + [read]
    monkeypatch.setenv("ITSX_KEEP_TRACE", "1")
    monkeypatch.delenv("ITSX_ALLOW_CAPS", raising=False)
    orc.mr_fail_counts(True)
    res = _run_both(engine, mini_hmm_text, seqs)
    st = engine.stats()
    assert orc.mr_fail_counts(True) == [0] * 8                      # the oracle holds no such limit any more
    assert st["n_mr_overflow"] >= 1 and st["n_mr_failed"] == 0, what   # the read really leaves the fast kernel
    _compare(engine, res)
    d = engine.domains()
    assert ((d["flags"] & 1) == 1).sum() >= 1
    full, _ = _coords(engine, mini_hmm_text, seqs, "full")
    assert full[0][3][-1] == 1                                       # the read has its entry
    monkeypatch.delenv("ITSX_KEEP_TRACE")
    lazy, st2 = _coords(engine, mini_hmm_text, seqs, "lazy")
    assert st2["lazy"] == 1 and all(np.array_equal(x, y) for a, b in zip(full, lazy) for x, y in zip(a, b))


@pytest.mark.parametrize("copies", [8, 12, 40])
def test_any_number_of_envelopes_per_pair(engine, mini_hmm_text, monkeypatch, copies):
    """a concatemer of `copies` copies of a profile's consensus: every envelope is kept (8 fit the pair's slots, the rest go through
    the overflow list), the rows equal the oracle's, and the argmax works with domain indices past 16"""
    from test_gpu_parity import _run_both
    cons3 = synth.consensus_motifs(mini_hmm_text, "3_")
    rng = np.random.default_rng(9)
    tail = "".join(rng.choice(list("ACGT"), 40))
    blob, offs = synth.make_reads(mini_hmm_text, 10, seed=5)
    seqs = synth.to_strings(blob, offs) + [cons3[0] * copies + tail]
    monkeypatch.setenv("ITSX_KEEP_TRACE", "1")
    res = _run_both(engine, mini_hmm_text, seqs)
    _compare(engine, res)
    st = engine.stats()
    d = engine.domains()
    assert d["ndom"].max() >= copies and (st["n_domain_overflow"] >= 1) == (copies > 8)
    monkeypatch.delenv("ITSX_KEEP_TRACE")
    full, _ = _coords(engine, mini_hmm_text, seqs, "full")
    lazy, _ = _coords(engine, mini_hmm_text, seqs, "lazy")
    assert all(np.array_equal(x, y) for a, b in zip(full, lazy) for x, y in zip(a, b))
    # the oracle's argmax on the same rows
    us, ue, ut, ui = res.positions("3_", "4_")
    _, _, uq = engine.get_derep()
    assert np.array_equal(full[0][0], us[uq]) and np.array_equal(full[0][1], ue[uq]) and np.array_equal(full[0][3], ui[uq])
