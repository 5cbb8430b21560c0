"""Refuse, don't cap.  hmmsearch has no limit on the envelopes of a target or on the bookkeeping of a region's traceback
ensemble; this engine has (8 envelopes per (representative, profile); per multidomain region at most 8 domains in one sampled
path, 512 distinct sampled tuples, 4 envelopes).  A search that runs into one of them would differ from the reference's
result on exactly those reads, so `itsx_search` fails with ITSX_E_UNSUPPORTED and says which limit and how often;
ITSX_ALLOW_CAPS=1 accepts the documented capped behaviour (the first 8 envelopes; an overrun region kept as ONE envelope,
flagged), which the oracle mirrors.  One adversarial read per limit, each found by a random search over tandem partial
copies of a profile's consensus (concatemer-like reads).  The remaining two constants cannot bind (k_api.h: 32 clusters of
>= 50 sampled domains each out of <= 1 600; 1 600 samples = 200 paths x 8 domains).  `pytest -m gpu`."""
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


def _search(engine, hmm, seqs):
    engine.load_profiles(text=hmm)
    engine.set_reads(seqs)
    engine.derep()
    engine.search()


@pytest.mark.parametrize("read,kind,what", [(READ_PATH_DOMAINS, 2, "more than 8 domains in one sampled path"),
                                            (READ_TUPLES, 4, "more than 512 distinct sampled tuples"),
                                            (READ_ENVELOPES, 7, "more than 4 envelopes in one region")])
def test_ensemble_limit_is_refused_and_the_allowed_fallback_equals_the_oracle(engine, mini_hmm_text, monkeypatch, read, kind, what):
    from itsxpress_amd import EngineError
    # an ordinary companion read: the refusal is per call, whatever else the call holds
    blob, offs = synth.make_reads(mini_hmm_text, 20, seed=3)
    seqs = synth.to_strings(blob, offs) + [read]
    # the oracle confirms that THIS limit is the one the read trips
    orc.mr_fail_counts(True)
    c, o = orc.digitize([read])
    orc.SearchResult(orc.HmmSet(text=mini_hmm_text), c, o, threads=1, keep_trace=0)
    fails = orc.mr_fail_counts(True)
    assert fails[kind] >= 1
    monkeypatch.delenv("ITSX_ALLOW_CAPS", raising=False)
    with pytest.raises(EngineError) as ei:
        _search(engine, mini_hmm_text, seqs)
    assert ei.value.code == -5 and what in str(ei.value) and "ITSX_ALLOW_CAPS" in str(ei.value)
    with pytest.raises(EngineError):
        engine.finalize()                                  # no result to finalize after a refused search
    # explicitly allowed: the region is kept as one envelope (flagged), like the oracle's mirror of the limit
    monkeypatch.setenv("ITSX_ALLOW_CAPS", "1")
    from test_gpu_parity import _run_both
    monkeypatch.setenv("ITSX_KEEP_TRACE", "1")
    res = _run_both(engine, mini_hmm_text, seqs)
    st = engine.stats()
    ofails = orc.mr_fail_counts(True)
    assert st["n_mr_fail_kind"][kind] >= 1 and st["n_mr_failed"] == sum(st["n_mr_fail_kind"])
    assert st["n_mr_fail_kind"] == ofails
    if st["n_domain_overflow"] == 0:                       # (the oracle keeps up to 64 envelopes per pair, the engine 8)
        _compare(engine, res)
    d = engine.domains()
    assert ((d["flags"] & 1) == 1).sum() >= 1
    start, stop, tlen, ind = engine.trim_coords("3_", "4_")
    assert ind[-1] == 1                                     # the read keeps its entry: the region did not vanish


def test_more_than_eight_envelopes_per_pair_are_refused(engine, mini_hmm_text, monkeypatch):
    from itsxpress_amd import EngineError
    cons3 = synth.consensus_motifs(mini_hmm_text, "3_")
    rng = np.random.default_rng(9)
    tail = "".join(rng.choice(list("ACGT"), 40))
    monkeypatch.delenv("ITSX_ALLOW_CAPS", raising=False)
    with pytest.raises(EngineError) as ei:
        _search(engine, mini_hmm_text, [cons3[0] * 12 + tail])
    assert ei.value.code == -5 and "more than 8 envelopes" in str(ei.value)
    # eight copies are fine
    _search(engine, mini_hmm_text, [cons3[0] * 8 + tail])
    engine.finalize()
    assert engine.stats()["n_domain_overflow"] == 0 and engine.domains()["ndom"].max() == 8
    # allowed: the first 8 are kept and the rest counted, per read too
    monkeypatch.setenv("ITSX_ALLOW_CAPS", "1")
    _search(engine, mini_hmm_text, [cons3[0] * 12 + tail])
    engine.finalize()
    st = engine.stats()
    assert st["n_domain_overflow"] >= 1 and engine.domains()["ndom"].max() == 8
    engine.trim_coords("3_", "4_")
    assert engine.stats()["n_reads_region_cap"] == 1
