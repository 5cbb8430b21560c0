"""The exact pin of the HMM half (SURVEY 7 hard part 1b, 8c P4-P6) -- armed the moment `F.hmm` exists.

The reference pins Fungi / ITS2 results: tests/test_main_pytest.py:32-46 (two literal domtbl rows, 137 keys), :68-161
(226 reads, 42 637 bases) and :378-397 (t2_r1.fq / t2_r2.fq byte for byte), from which SURVEY 8c-P5 derived the 226
(start, stop, tlen) triples of tests/golden/fungi_its2_coords.tsv.  The Fungi model file itself
(`itsxpress/ITSx_db/HMMs/F.hmm`) is missing from the reference mount, so until now these numbers could only be
approached with stand-in models (tests/test_oracle_cpu.py).  Here: when `ITSx_db/HMMs/F.hmm` is found through
$ITSXPRESS_DB_DIR or an installed `itsxpress` package, the CPU oracle (`-m "not gpu"`) and the HIP engine (`-m gpu`)
must each reproduce ALL of it EXACTLY; when it is absent the tests skip, naming the file.  Nothing of the reference's
Python is needed for this -- only the model file.
"""
import gzip
import json
import os

import numpy as np
import pytest

from itsxpress_amd.definitions import hmm_path

F_HMM = hmm_path("Fungi")
need_f = pytest.mark.skipif(F_HMM is None, reason="ITSx_db/HMMs/F.hmm (the Fungi profile set) not found: set ITSXPRESS_DB_DIR to a "
                            "directory that holds ITSx_db/HMMs/F.hmm to arm the exact pin of the 226 golden trim coordinates")

LIT1 = "M02696:28:000000000-ATWK5:1:1101:19331:3209"      # tests/test_main_pytest.py:36-41
LIT2 = "M02696:28:000000000-ATWK5:1:1101:23011:4341"      # tests/test_main_pytest.py:43-45


def _golden(gold):
    rows = [ln.split("\t") for ln in open(os.path.join(gold, "fungi_its2_coords.tsv")).read().strip().split("\n")[1:]]
    assert len(rows) == 226
    return rows


def _runtime_hmm(tmp_path):
    """create_runtime_hmm("Fungi", "ITS2") exactly as main.py:542 calls it"""
    from itsxpress_amd.main import create_runtime_hmm
    d = tmp_path / "rt"
    d.mkdir(exist_ok=True)
    path = create_runtime_hmm("Fungi", "ITS2", str(d))
    text = open(path).read()
    assert text.count("\nNAME  ") + text.startswith("NAME  ") > 0 or "NAME  " in text
    return path, text


def _check_ddict(ddict, rows):
    """the reference's literal rows, its key count, and the 226 triples through get_position's arithmetic"""
    assert ddict[LIT1] == {"tlen": 341, "right": {"score": 59.1, "to_pos": 326, "from_pos": 282},
                           "left": {"score": 52.2, "to_pos": 128, "from_pos": 84}}
    assert ddict[LIT2] == {"right": {"score": 34.0, "to_pos": 370, "from_pos": 327}, "tlen": 385}
    assert len(ddict) == 137
    bad = []
    for rid, rep, a, b, t in rows:
        e = ddict[rep]
        got = (e["left"]["to_pos"], e["right"]["from_pos"] - 1, e["tlen"])
        if got != (int(a), int(b), int(t)):
            bad.append((rid, got, (a, b, t)))
    assert not bad, bad[:5]


@need_f
def test_oracle_reproduces_the_fungi_goldens_exactly(fixture_reads, gold, tmp_path):
    import orc
    from itsxpress_amd import ItsPosition
    from itsxpress_amd.trim import write_trimmed_fastq
    from itsxpress_amd.engine import read_fastx
    names, seqs = fixture_reads
    _, text = _runtime_hmm(tmp_path)
    codes, offs = orc.digitize(seqs)
    nc, rep_of, _ = orc.derep(codes, offs)
    assert nc == 137
    seeds = [i for i in range(len(seqs)) if rep_of[i] == i]
    c2, o2 = orc.digitize([seqs[i] for i in seeds])
    hs = orc.HmmSet(text=text)
    res = orc.SearchResult(hs, c2, o2, threads=8)
    # the reported rows as domtbl text, read back by the mirror of ItsPosition (columns 0, 2, 3, 13, 19, 20)
    p = tmp_path / "domtbl.txt"
    with open(p, "w") as f:
        for prof in range(len(hs.names)):
            for d in res.domains[(res.domains["prof"] == prof) & (res.domains["dom_reported"] == 1)]:
                f.write("%s - %d %s - 45 1e-9 %.1f 0.0 1 1 1e-9 1e-9 %6.1f 0.0 1 45 %d %d %d %d 0.9 -\n" % (
                    names[seeds[int(d["seq"])]], d["tlen"], hs.names[prof], d["seq_score"], d["bitscore"],
                    d["ienv"], d["jenv"], d["ienv"], d["jenv"]))
    rows = _golden(gold)
    _check_ddict(ItsPosition(str(p), "ITS2").ddict, rows)
    # 226 reads / 42 637 bases through the native writer (tests/test_main_pytest.py:95-96)
    start, stop, tlen, ind = res.positions("3_", "4_")
    uniq = np.cumsum(np.asarray(rep_of) == np.arange(len(seqs))) - 1
    uo = uniq[np.asarray(rep_of)]
    fq_names, _ = read_fastx(os.path.join(gold, "seq.fq.gz"))
    assert fq_names == names
    out = str(tmp_path / "trimmed.fastq")
    assert write_trimmed_fastq(os.path.join(gold, "seq.fq.gz"), out, start[uo].astype(np.int32), stop[uo].astype(np.int32)) == (226, 42637)


@need_f
@pytest.mark.gpu
def test_engine_reproduces_the_fungi_goldens_exactly(gold, tmp_path):
    """the whole mirrored path on the device: deduplicate -> create_runtime_hmm(Fungi, ITS2) -> _search -> ItsPosition / Dedup ->
    the paired writer, against every number the reference's tests hold"""
    from itsxpress_amd import Dedup, ItsPosition, SeqSampleNotPaired
    raw = []
    for fn in ("4774-1-MSITS3_R1.fastq", "4774-1-MSITS3_R2.fastq"):
        p = tmp_path / fn
        with gzip.open(os.path.join(gold, fn + ".gz"), "rt") as f:
            p.write_text(f.read())
        raw.append(str(p))
    seq = tmp_path / "seq.fq"
    with gzip.open(os.path.join(gold, "seq.fq.gz"), "rt") as f:
        seq.write_text(f.read())
    work = tmp_path / "work"
    work.mkdir()
    s = SeqSampleNotPaired(str(seq), str(work))
    s.deduplicate(threads=1)
    assert open(s.uc_file).read() == open(os.path.join(gold, "fixture_uc.txt")).read()
    hmmfile, _ = _runtime_hmm(tmp_path)
    s._search(hmmfile=hmmfile, threads=1)
    rows = _golden(gold)
    ip = ItsPosition(s.dom_file, "ITS2")
    _check_ddict(ip.ddict, rows)
    dd = Dedup(s.uc_file, s.rep_file, s.seq_file, fastq=raw[0], fastq2=raw[1])
    assert dd.matchdict == json.load(open(os.path.join(gold, "matchdict.json")))
    # single-end: 226 reads, 42 637 bases (tests/test_main_pytest.py:68-161)
    out = str(tmp_path / "trimmed.fastq")
    dd.create_trimmed_seqs(out, gzipped=False, zstd_file=False, itspos=ip)
    recs = open(out).read().split("\n")
    assert len(recs) == 226 * 4 + 1 and sum(len(recs[i]) for i in range(1, 226 * 4, 4)) == 42637
    # paired: the reference's byte-compared goldens (tests/test_main_pytest.py:378-397)
    o1, o2 = str(tmp_path / "t2_r1.fq"), str(tmp_path / "t2_r2.fq")
    dd.create_paired_trimmed_seqs(o1, o2, gzipped=False, zstd_file=False, itspos=ip)
    for o, g in ((o1, "t2_r1.fq.gz"), (o2, "t2_r2.fq.gz")):
        with gzip.open(os.path.join(gold, g), "rt") as f:
            assert open(o).read() == f.read()
    # and the array path gives the same coordinates without the text round trip
    start, stop, tlen, ind = s.trim_coordinates("ITS2")
    gold_by_read = {rid: (int(a), int(b), int(t)) for rid, rep, a, b, t in rows}
    names = s.engine.read_names()
    for i, n in enumerate(names):
        if n in gold_by_read:
            assert (int(start[i]), int(stop[i]), int(tlen[i])) == gold_by_read[n], n


def test_pin_is_gated_on_the_file_and_create_runtime_hmm_says_so(tmp_path, monkeypatch, caplog):
    """without F.hmm: the helper returns None, create_runtime_hmm writes an empty selection like the reference does
    (main.py:214-215) -- and logs the missing file instead of staying silent"""
    import importlib
    import logging
    from itsxpress_amd import definitions
    monkeypatch.setenv("ITSXPRESS_DB_DIR", str(tmp_path))
    importlib.reload(definitions)
    try:
        assert definitions.hmm_path("Fungi") is None
        from itsxpress_amd.main import create_runtime_hmm
        with caplog.at_level(logging.WARNING):
            out = create_runtime_hmm("Fungi", "ITS2", str(tmp_path))
        assert open(out).read() == ""
        assert "F.hmm" in caplog.text and "not found" in caplog.text
        # a supplied file is found (any HMMER3/f text will do for the plumbing)
        d = tmp_path / "ITSx_db" / "HMMs"
        d.mkdir(parents=True)
        with open(os.path.join(os.path.dirname(__file__), "golden", "mini.hmm")) as f:
            (d / "F.hmm").write_text(f.read())
        assert definitions.hmm_path("Fungi") == str(d / "F.hmm")
        out = create_runtime_hmm("Fungi", "ITS2", str(tmp_path))
        assert "NAME  " in open(out).read()
    finally:
        monkeypatch.delenv("ITSXPRESS_DB_DIR")
        importlib.reload(definitions)
