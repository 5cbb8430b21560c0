"""ITSXPRESS_DOMTBL=winners (itsx_set_kept_rows): the file-compatible outputs with the lazy search behind them.  domtbl.txt then holds,
per target and side, the row ItsPosition.parse (itsxpress/SeqSample.py:400-461) ends up with.  Every listed row must be a row of the
full table, field for field; none twice (a top-up / completion round evaluates pairs a second time); one per target and prefix; and the
parser must read the SAME dictionary out of the thinned file as out of the full one, for every region.  `pytest -m gpu`."""
import os

import numpy as np
import pytest

import synth
from test_gpu_parity import _its2_subset

pytestmark = pytest.mark.gpu

REGIONS = ("ITS2", "ITS1", "ALL")


def _damaged(mini_hmm_text, n, seed):
    """reads of which many keep only rows near the Forward filter's threshold (rows whose reporting depends on domZ)"""
    rng = np.random.default_rng(37)
    blob, offs = synth.make_reads(mini_hmm_text, n, seed=seed, fixed_len=0, len_range=(200, 520))
    seqs = []
    for s in synth.to_strings(blob, offs):
        if rng.random() < 0.6:
            s = list(s)
            for q in rng.choice(len(s), int(len(s) * rng.uniform(0.3, 0.42)), replace=False):
                s[q] = str(rng.choice(list("ACGT")))
            s = "".join(s)
        seqs.append(s)
    return seqs


def _run(engine, hmm, seqs, mode, tmp_path, tag):
    from itsxpress_amd import ItsPosition
    engine.set_rows_mode(mode)
    engine.load_profiles(text=hmm)
    engine.set_reads(seqs)
    engine.derep()
    engine.search()
    engine.finalize()
    engine.set_kept_rows(mode != "full")
    rows = engine.domains()
    p = tmp_path / ("domtbl_%s.txt" % tag)
    engine.write_domtbl(str(p))
    dd = {r: ItsPosition(str(p), r).ddict for r in REGIONS}
    return rows, p, dd, engine.stats()


@pytest.mark.parametrize("hooks", [None, "topup", "count"])
def test_kept_rows_are_rows_of_the_full_table_and_parse_alike(engine, mini_hmm_text, t_hmm_text, tmp_path, monkeypatch, hooks):
    from itsxpress_amd import EngineError
    seqs = _damaged(mini_hmm_text, 2500, 75)
    hmm = mini_hmm_text + _its2_subset(t_hmm_text, 25, 25)            # 1_ 2_ 3_ 4_ profiles, families of near-identical ones
    if hooks:                                                        # undecided rows -> pairs evaluated a second time (duplicates to remove)
        monkeypatch.setenv("ITSX_LAZY_ZUB_SCALE", "500000")
        monkeypatch.setenv("ITSX_LAZY_TOPUP", "1" if hooks == "topup" else "0")
    try:
        full, pf, dfull, _ = _run(engine, hmm, seqs, "full", tmp_path, "full")
        kept, pk, dkept, st = _run(engine, hmm, seqs, "lazy", tmp_path, "kept")
        assert st["lazy"] == 1 and st["n_lazy_reruns"] == 0
        if hooks:
            assert st["n_lazy_pending"] > 0 and (st["n_lazy_topup"] > 0 if hooks == "topup" else st["n_lazy_completed"] > 0)
        assert 0 < len(kept) < 0.25 * len(full)
        # every kept row is the full table's row of that (profile, target, domain), field for field (domZ-independent fields and the decision)
        key = lambda a: (a["prof"].astype(np.int64) << 44) | (a["rep"] << 16) | a["dom_idx"].astype(np.int64)
        kf, kk = key(full), key(kept)
        assert len(np.unique(kk)) == len(kk), "a (profile, target, domain) is listed twice"
        assert (kept["dom_reported"] == 1).all()
        names = engine.profile_names()
        grp = kept["rep"] * 8 + np.array([int(names[p][0]) for p in kept["prof"]])          # (prefixes 1_ 2_ 3_ 4_)
        assert len(np.unique(grp)) == len(grp), "more than one row of a target and prefix"
        order = np.argsort(kf)
        pos = np.searchsorted(kf[order], kk)
        assert (pos < len(kf)).all() and (kf[order][pos] == kk).all()
        twin = full[order][pos]
        for f in ("tlen", "ienv", "jenv", "ndom", "envsc", "domcorrection", "dombias", "bitscore", "lnP", "seq_score", "seq_bias", "seq_reported", "dom_reported"):
            assert np.array_equal(twin[f], kept[f]), f
        # domtblout order in the file, and the parser's dictionary: the full table's, for every region
        assert np.array_equal(kk, np.sort(kk))
        for r in REGIONS:
            assert dkept[r] == dfull[r], r
            assert len(dfull[r]) > 100
        nf = sum(1 for ln in open(pf) if not ln.startswith("#"))
        nk = sum(1 for ln in open(pk) if not ln.startswith("#"))
        assert 0 < nk < 0.25 * nf
        # without the request a thinned table is still refused
        engine.set_kept_rows(False)
        with pytest.raises(EngineError):
            engine.domains()
    finally:
        engine.set_kept_rows(False)
        engine.set_rows_mode(None)


def test_mirror_writes_the_winners_table(fixture_reads, mini_hmm_text, tmp_path, monkeypatch):
    """SeqSample._search with ITSXPRESS_DOMTBL=winners: uc.txt / rep.fa / domtbl.txt as files, the parsers' dictionaries and the trimmed
    coordinates those of the default (full-table) mode."""
    from itsxpress_amd import Dedup, ItsPosition, SeqSampleNotPaired
    names, seqs = fixture_reads
    fq = tmp_path / "seq.fq"
    with open(fq, "w") as f:
        for n, s in zip(names, seqs):
            f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
    hmm = tmp_path / "mini.hmm"
    hmm.write_text(mini_hmm_text)
    res = {}
    for mode in ("full", "winners"):
        d = tmp_path / mode
        d.mkdir()
        if mode == "winners":
            monkeypatch.setenv("ITSXPRESS_DOMTBL", "winners")
        s = SeqSampleNotPaired(str(fq), str(d))
        try:
            s.deduplicate(threads=1)
            s._search(hmmfile=str(hmm), threads=1)
            assert os.path.exists(s.uc_file) and os.path.exists(s.rep_file) and os.path.exists(s.dom_file) and type(s.dom_file) is str
            dd = Dedup(s.uc_file, s.rep_file, s.seq_file)
            res[mode] = (dd.matchdict, {r: ItsPosition(s.dom_file, r).ddict for r in REGIONS},
                         sum(1 for ln in open(s.dom_file) if not ln.startswith("#")), s.engine.stats()["lazy"])
        finally:
            s.engine.close()
    assert res["full"][3] == 0 and res["winners"][3] == 1
    assert res["full"][0] == res["winners"][0] and len(res["full"][0]) == 227
    assert res["full"][1] == res["winners"][1] and len(res["full"][1]["ITS2"]) > 50
    assert 0 < res["winners"][2] < res["full"][2]


@pytest.mark.parametrize("how", ["two_workers", "streamed"])
def test_winners_table_composed_from_several_contexts(tmp_path, t_hmm_text, monkeypatch, how):
    """ITSXPRESS_GPUS=2 (two workers on this GPU) and the streamed engine (file-order chunks) compose the kept rows of their contexts:
    the parser's dictionary and the trimmed FASTQ are those of one engine's full table."""
    import test_gpu_multi as M
    tmp = str(tmp_path)
    hmm = M._its2(tmp, t_hmm_text)
    fq = os.path.join(tmp, "synth.fq")
    M._fastq(fq, t_hmm_text, 3000, 4242)
    try:
        monkeypatch.delenv("ITSXPRESS_DOMTBL", raising=False)
        monkeypatch.setenv("ITSXPRESS_STREAM", "0")
        one, out1, c1, pos1, dd1 = M._run(fq, os.path.join(tmp, "one"), hmm, 1, False, monkeypatch)
        monkeypatch.setenv("ITSXPRESS_DOMTBL", "winners")
        if how == "streamed":
            monkeypatch.setenv("ITSXPRESS_STREAM", "1")
            monkeypatch.setenv("ITSX_STREAM_CHUNK_MB", "0.4")
        two, out2, c2, pos2, dd2 = M._run(fq, os.path.join(tmp, "two"), hmm, 2 if how == "two_workers" else 1, False, monkeypatch)
        assert type(two.dom_file) is str and os.path.exists(two.dom_file)
        if how == "streamed":
            assert type(two.engine).__name__ == "StreamEngine" and two.engine.world > 2
        assert pos1.ddict == pos2.ddict and len(pos1.ddict) > 1000
        assert dd1.matchdict == dd2.matchdict
        assert out1 == out2 and all(np.array_equal(a, b) for a, b in zip(c1, c2))
        n1 = sum(1 for ln in open(one.dom_file) if not ln.startswith("#"))
        n2 = sum(1 for ln in open(two.dom_file) if not ln.startswith("#"))
        assert 0 < n2 < 0.25 * n1
    finally:
        while M._OPEN:
            s = M._OPEN.pop()
            if getattr(s, "_engine", None) is not None:
                s._engine.close()


def test_sample_batch_writes_winners_tables(engine, fixture_reads, t_hmm_text, mini_hmm_text, tmp_path, monkeypatch):
    """SampleBatch (one engine pass over many samples, per-sample domZ): every sample's domtbl.txt in the winners mode parses to the
    dictionary of its default-mode file."""
    from itsxpress_amd import ItsPosition, SeqSampleNotPaired
    from itsxpress_amd.batch import SampleBatch
    names, seqs = fixture_reads
    blob, offs = synth.make_reads(t_hmm_text, 600, seed=33)
    syn = synth.to_strings(blob, offs)
    parts = [(names[:120], seqs[:120]), (names[120:], seqs[120:]), (["s%05d" % i for i in range(len(syn))], syn)]
    hmm = tmp_path / "its2.hmm"
    hmm.write_text(mini_hmm_text + _its2_subset(t_hmm_text, 12, 12))
    fqs = []
    for k, (nm, sq) in enumerate(parts):
        fq = tmp_path / ("sample%d.fq" % k)
        with open(fq, "w") as f:
            for n, s in zip(nm, sq):
                f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
        fqs.append(str(fq))
    got = {}
    try:
        for mode in ("full", "winners"):
            d = tmp_path / mode
            os.makedirs(d)
            if mode == "winners":
                monkeypatch.setenv("ITSXPRESS_DOMTBL", "winners")
            objs = [SeqSampleNotPaired(fq, str(d)) for fq in fqs]
            b = SampleBatch(objs, engine=engine)
            b.deduplicate(threads=1)
            b._search(hmmfile=str(hmm), threads=1)
            got[mode] = [({r: ItsPosition(s.dom_file, r).ddict for r in REGIONS}, sum(1 for ln in open(s.dom_file) if not ln.startswith("#"))) for s in objs]
    finally:
        engine.set_kept_rows(False)
        engine.set_rows_mode(None)
    for k in range(len(parts)):
        assert got["full"][k][0] == got["winners"][k][0], k
        assert 0 < got["winners"][k][1] < got["full"][k][1]
    assert len(got["full"][2][0]["ITS2"]) > 300
