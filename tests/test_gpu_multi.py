"""ITSXPRESS_GPUS=N behind the mirror (itsxpress_amd/multi.py): two worker processes -- both on GPU 0 here, a one-GPU box -- must
give what one engine gives on the whole sample: uc.txt, rep.fa and domtbl.txt byte for byte, per-read coordinates, and, in
arrays mode, the trimmed FASTQ.  Reference call sequence: itsxpress/main.py:534-554, 626-638.  `pytest -m gpu`."""
import gzip
import importlib
import os

import numpy as np
import pytest

import synth
from conftest import GOLD

pytestmark = pytest.mark.gpu
S = importlib.import_module("itsxpress_amd.SeqSample")


def _its2(tmp, t_hmm_text):
    from bench import its2_profiles
    p = os.path.join(tmp, "its2.hmm")
    with open(p, "w") as f:
        f.write(its2_profiles(t_hmm_text))
    return p


def _fastq(path, t_hmm_text, n, seed):
    """reads with duplicates across the two halves of the file and reverse-complemented copies of earlier reads"""
    blob, offs = synth.make_reads(t_hmm_text, n, config=3, seed=seed, fixed_len=0, len_range=(300, 520), rc_rate=0.2)
    seqs = synth.to_strings(blob, offs)
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write("@read%05d extra words\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    return seqs


def _run(fq, tmp, hmm, gpus, fast, monkeypatch):
    os.makedirs(tmp, exist_ok=True)
    monkeypatch.setenv("ITSXPRESS_GPUS", str(gpus))
    monkeypatch.setenv("ITSXPRESS_GPU_IDS", ",".join(["0"] * gpus))
    monkeypatch.setenv("ITSXPRESS_ARRAYS", "1" if fast else "0")
    sobj = S.SeqSampleNotPaired(fastq=fq, tempdir=tmp)
    _OPEN.append(sobj)
    sobj.deduplicate(threads=1)
    sobj._search(hmmfile=hmm, threads=1)
    its_pos = S.ItsPosition(domtable=sobj.dom_file, region="ITS2")
    dedup_obj = S.Dedup(uc_file=sobj.uc_file, rep_file=sobj.rep_file, seq_file=sobj.seq_file, fastq=sobj.r1, fastq2=sobj.fastq2)
    out = os.path.join(tmp, "trimmed.fq")
    dedup_obj.create_trimmed_seqs(out, gzipped=False, zstd_file=False, itspos=its_pos, wri_file=True, tempdir=tmp)
    coords = sobj.trim_coordinates("ITS2")
    return sobj, open(out, "rb").read(), [np.asarray(c).copy() for c in coords], its_pos, dedup_obj


_OPEN = []


@pytest.fixture(autouse=True)
def _close_engines():
    yield
    while _OPEN:
        s = _OPEN.pop()
        if getattr(s, "_engine", None) is not None:
            s._engine.close()


@pytest.mark.parametrize("source", ["fixture", "synthetic"])
def test_two_workers_equal_one_engine_file_for_file(tmp_path, t_hmm_text, monkeypatch, source):
    tmp = str(tmp_path)
    hmm = _its2(tmp, t_hmm_text)
    if source == "fixture":
        fq = os.path.join(tmp, "seq.fq")
        with gzip.open(os.path.join(GOLD, "seq.fq.gz"), "rb") as f, open(fq, "wb") as g:
            g.write(f.read())
    else:
        fq = os.path.join(tmp, "synth.fq")
        _fastq(fq, t_hmm_text, 3000, 4242)
    one, out1, c1, pos1, dd1 = _run(fq, os.path.join(tmp, "one"), hmm, 1, False, monkeypatch)
    two, out2, c2, pos2, dd2 = _run(fq, os.path.join(tmp, "two"), hmm, 2, False, monkeypatch)
    for name in ("uc.txt", "rep.fa", "domtbl.txt"):
        a = open(os.path.join(tmp, "one", name), "rb").read()
        b = open(os.path.join(tmp, "two", name), "rb").read()
        assert len(a) > 100 and a == b, name
    assert out1 == out2 and len(out1) > 1000
    assert all(np.array_equal(x, y) for x, y in zip(c1, c2))
    assert pos1.ddict == pos2.ddict and dd1.matchdict == dd2.matchdict
    # arrays mode (lazy domain stage, nothing written): one engine and two workers give the same trimmed file as the text path
    f1, outf1, cf1, posf1, ddf1 = _run(fq, os.path.join(tmp, "fast1"), hmm, 1, True, monkeypatch)
    f2, outf2, cf2, posf2, ddf2 = _run(fq, os.path.join(tmp, "fast2"), hmm, 2, True, monkeypatch)
    assert outf1 == out1 and outf2 == out1
    assert all(np.array_equal(x, y) for x, y in zip(c1, cf1)) and all(np.array_equal(x, y) for x, y in zip(c1, cf2))
    assert not os.path.exists(os.path.join(tmp, "fast2", "domtbl.txt")) and not os.path.exists(os.path.join(tmp, "fast2", "uc.txt"))
    assert isinstance(f2.dom_file, S.EngineTable) and isinstance(f2.uc_file, S.EngineTable)
    # the reference-shaped dicts built from the arrays agree with the parsed files on what get_position reads
    assert ddf2.matchdict == dd1.matchdict
    for k, e in pos1.ddict.items():
        if "left" in e or "right" in e:
            assert posf2.get_position(k) == pos1.get_position(k), k
    assert set(k for k, e in pos1.ddict.items()) == set(posf2.ddict.keys())


def test_three_workers_with_an_empty_shard(tmp_path, t_hmm_text, monkeypatch):
    """more workers than reads in a shard's worth: 2 reads over 3 workers (one shard is empty)"""
    tmp = str(tmp_path)
    hmm = _its2(tmp, t_hmm_text)
    fq = os.path.join(tmp, "tiny.fq")
    _fastq(fq, t_hmm_text, 2, 7)
    one, out1, c1, _, _ = _run(fq, os.path.join(tmp, "one"), hmm, 1, False, monkeypatch)
    three, out3, c3, _, _ = _run(fq, os.path.join(tmp, "three"), hmm, 3, False, monkeypatch)
    assert out1 == out3 and all(np.array_equal(x, y) for x, y in zip(c1, c3))
    for name in ("uc.txt", "rep.fa", "domtbl.txt"):
        assert open(os.path.join(tmp, "one", name), "rb").read() == open(os.path.join(tmp, "three", name), "rb").read()


def test_paired_sample_merge_and_orientation_over_two_workers(tmp_path, t_hmm_text, monkeypatch):
    """round 5: merging and orientation are sharded by record range like the reads (R2 cut at R1's record counts: csrc/shard_host.cpp).
    Two workers == one engine on seq.fq, the trimmed pairs, the single merged output and oriented.fq -- in file mode and in arrays mode"""
    tmp = str(tmp_path)
    hmm = _its2(tmp, t_hmm_text)
    r1, r2 = os.path.join(GOLD, "4774-1-MSITS3_R1.fastq.gz"), os.path.join(GOLD, "4774-1-MSITS3_R2.fastq.gz")
    res = {}
    for gpus, fast in ((1, False), (2, False), (2, True), (3, True)):
        monkeypatch.setenv("ITSXPRESS_GPUS", str(gpus))
        monkeypatch.setenv("ITSXPRESS_GPU_IDS", ",".join(["0"] * gpus))
        monkeypatch.setenv("ITSXPRESS_ARRAYS", "1" if fast else "0")
        monkeypatch.setenv("ITSXPRESS_STREAM", "0")
        d = os.path.join(tmp, "p%d%d" % (gpus, fast))
        s = S.SeqSamplePairedNotInterleaved(fastq=r1, tempdir=d, fastq2=r2)
        _OPEN.append(s)
        s._merge_reads(threads=1, stagger=False)
        assert os.path.exists(os.path.join(d, "seq.fq")) == (not fast)
        s.deduplicate(threads=1)
        s._search(hmmfile=hmm, threads=1)
        pos = S.ItsPosition(domtable=s.dom_file, region="ITS2")
        dd = S.Dedup(uc_file=s.uc_file, rep_file=s.rep_file, seq_file=s.seq_file, fastq=s.r1, fastq2=s.fastq2)
        o1, o2, om = os.path.join(d, "o1.fq"), os.path.join(d, "o2.fq"), os.path.join(d, "om.fq")
        dd.create_paired_trimmed_seqs(o1, o2, gzipped=False, zstd_file=False, itspos=pos, wri_file=True)
        dd.create_trimmed_seqs(om, gzipped=False, zstd_file=False, itspos=pos, wri_file=True, tempdir=d)
        res[(gpus, fast)] = (open(o1, "rb").read(), open(o2, "rb").read(), open(om, "rb").read(), open(os.path.join(d, "seq.fq"), "rb").read(),
                             [np.asarray(c).copy() for c in s.trim_coordinates("ITS2")])
        s._engine.close()
        _OPEN.pop()
    ref = res[(1, False)]
    assert len(ref[0]) > 1000 and len(ref[2]) > 1000 and ref[3].count(b"\n") == 4 * 236
    for k, v in res.items():
        assert v[:4] == ref[:4], k
        assert all(np.array_equal(x, y) for x, y in zip(v[4], ref[4])), k
    # orientation (SeqSample.orient_reads, the --trim-ccs front end): oriented.fq of two workers == one engine's
    monkeypatch.setenv("ITSXPRESS_DB_DIR", GOLD)
    import itsxpress_amd.definitions as D
    importlib.reload(D)
    outs = []
    for gpus in (1, 2):
        monkeypatch.setenv("ITSXPRESS_GPUS", str(gpus))
        monkeypatch.setenv("ITSXPRESS_GPU_IDS", ",".join(["0"] * gpus))
        d = os.path.join(tmp, "o%d" % gpus)
        s = S.SeqSampleNotPaired(fastq=os.path.join(GOLD, "seq.fq.gz"), tempdir=d)
        _OPEN.append(s)
        s.orient_reads(threads=1)
        outs.append(open(s.seq_file, "rb").read())
        s._engine.close()
        _OPEN.pop()
    assert outs[0] == outs[1] and outs[0].count(b"\n") >= 800
