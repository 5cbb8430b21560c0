"""ITSX_COMPACT_ROWS=1 (the bench's and the array path's mode): per (representative, 2-character profile prefix) only the domain
rows that can still win ItsPosition's argmax -- whatever the dataset-wide domZ turns out to be after the all-reduce -- stay
resident; a chunk's full row table is scratch.  The per-read coordinates and the "sequence has a row" flag must be exactly
those of the uncompacted search: compared on every prefix pair the reference uses, on one chunk and many, with every row
forced "uncertain" and with as many rows "certain" as the data allow; a finalize that breaks the compaction's assumptions is
refused.  `pytest -m gpu`."""
import numpy as np
import pytest

import synth
from test_gpu_parity import _its2_subset

pytestmark = pytest.mark.gpu

PAIRS = [("3_", "4_"), ("1_", "4_"), ("1_", "2_")]


def _weak(seqs, rng, frac=0.3):
    """damage the motifs of some reads so that their best domains score near the thresholds (rows that are NOT certain)"""
    out = []
    for s in seqs:
        if rng.random() < frac:
            s = list(s)
            for p in rng.choice(len(s), int(len(s) * rng.uniform(0.04, 0.12)), replace=False):
                s[p] = str(rng.choice(list("ACGT")))
            s = "".join(s)
        out.append(s)
    return out


def _coords(engine, hmm, seqs, domE=10.0):
    engine.load_profiles(text=hmm)
    engine.set_reads(seqs)
    engine.derep()
    engine.search()
    engine.finalize(domE=domE)
    st = engine.stats()
    return [tuple(a.copy() for a in engine.trim_coords(l, r)) for l, r in PAIRS], st


def _same(a, b):
    return all(np.array_equal(x, y) for pa, pb in zip(a, b) for x, y in zip(pa, pb))


@pytest.mark.parametrize("chunk", [None, "37"])
def test_compacted_rows_give_the_same_coordinates(engine, mini_hmm_text, t_hmm_text, monkeypatch, chunk):
    from itsxpress_amd import EngineError
    rng = np.random.default_rng(17)
    blob, offs = synth.make_reads(mini_hmm_text, 1200, seed=71, fixed_len=0, len_range=(200, 520))
    seqs = _weak(synth.to_strings(blob, offs), rng)
    hmm = mini_hmm_text + _its2_subset(t_hmm_text, 25, 25)            # 1_ 2_ 3_ 4_ profiles, families of near-identical ones
    if chunk:
        monkeypatch.setenv("ITSX_CHUNK_UNIQUES", chunk)
    monkeypatch.delenv("ITSX_COMPACT_ROWS", raising=False)
    ref, st0 = _coords(engine, hmm, seqs)
    assert st0["n_rows_resident"] >= st0["n_domains"] > 5000
    assert sum(int(((c[0] >= 0) | (c[1] >= 0)).sum()) for c in ref) > 1500
    monkeypatch.setenv("ITSX_COMPACT_ROWS", "1")
    # (a) default assumptions (domZ <= 1e9, domE >= 0.01)
    got, st1 = _coords(engine, hmm, seqs)
    assert _same(ref, got)
    assert st1["n_domains"] == st0["n_domains"] and st1["n_rows_resident"] * 4 < st0["n_domains"]
    with pytest.raises(EngineError):
        engine.domains()                                    # the row table is gone: coordinates only
    # (b) nothing is certain: every row of a reported target stays, same answer
    monkeypatch.setenv("ITSX_COMPACT_ZMAX", "1e300")
    got, st2 = _coords(engine, hmm, seqs)
    assert _same(ref, got) and st2["n_rows_resident"] > st1["n_rows_resident"]
    # (c) as much as the data allow: Zmax = the number of targets, domE_min = the domE finalize gets
    monkeypatch.setenv("ITSX_COMPACT_ZMAX", str(st0["n_unique"]))
    monkeypatch.setenv("ITSX_COMPACT_DOME_MIN", "10")
    got, st3 = _coords(engine, hmm, seqs)
    assert _same(ref, got) and st3["n_rows_resident"] <= st1["n_rows_resident"]
    # a stricter domE than assumed, or more reported targets than assumed, cannot be served from the thinned rows
    with pytest.raises(EngineError):
        _coords(engine, hmm, seqs, domE=1.0)
    monkeypatch.setenv("ITSX_COMPACT_ZMAX", "1")
    with pytest.raises(EngineError):
        _coords(engine, hmm, seqs)
    # a different domE within the assumptions is served exactly
    monkeypatch.delenv("ITSX_COMPACT_ZMAX")
    monkeypatch.delenv("ITSX_COMPACT_DOME_MIN")
    monkeypatch.delenv("ITSX_COMPACT_ROWS")
    ref2, _ = _coords(engine, hmm, seqs, domE=0.05)
    monkeypatch.setenv("ITSX_COMPACT_ROWS", "1")
    got2, _ = _coords(engine, hmm, seqs, domE=0.05)
    assert _same(ref2, got2)


def test_compaction_survives_the_domz_exchange(engine, mini_hmm_text, monkeypatch):
    """multi-GPU: domZ is summed over the ranks between search and finalize; a larger domZ un-reports weak rows, and the
    compacted table must follow (set_domz plays the all-reduce)"""
    rng = np.random.default_rng(18)
    blob, offs = synth.make_reads(mini_hmm_text, 800, seed=72, fixed_len=0, len_range=(200, 420))
    seqs = _weak(synth.to_strings(blob, offs), rng, frac=0.5)
    out = []
    for compact in (False, True):
        if compact:
            monkeypatch.setenv("ITSX_COMPACT_ROWS", "1")
        engine.load_profiles(text=mini_hmm_text)
        engine.set_reads(seqs)
        engine.derep()
        engine.search()
        engine.set_domz(engine.get_domz() * 50000)          # as if 50 000 ranks' worth of targets had been reported
        engine.finalize()
        out.append([tuple(a.copy() for a in engine.trim_coords(l, r)) for l, r in PAIRS])
    assert _same(out[0], out[1])
