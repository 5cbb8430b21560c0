"""ITSXPRESS_STREAM=1 (itsxpress_amd/stream.py): one FASTQ cut into file-order chunks -- each chunk dereplicated and scored while the
rest of the file is still being inflated -- must give what one engine gives on the whole file: uc.txt, rep.fa, domtbl.txt byte for
byte, per-read coordinates and the trimmed FASTQ.  Reference call sequence: itsxpress/main.py:534-554, 626-638.  `pytest -m gpu`."""
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


def _fastq_gz(path, t_hmm_text, n, seed, damage=0.0):
    """duplicates all over the file (also reverse-complemented copies of earlier reads: rc_rate), qualities that start with '@';
    damage: that share of the reads gets 30-42 % of its bases replaced (rows near the thresholds)"""
    blob, offs = synth.make_reads(t_hmm_text, n, config=3, seed=seed, fixed_len=0, len_range=(300, 520), rc_rate=0.2)
    seqs = synth.to_strings(blob, offs)
    rng = np.random.default_rng(seed)
    if damage > 0:
        for i in range(len(seqs)):
            if rng.random() < damage:
                s = list(seqs[i])
                for q in rng.choice(len(s), int(len(s) * rng.uniform(0.3, 0.42)), replace=False):
                    s[q] = "ACGT"[int(rng.integers(0, 4))]
                seqs[i] = "".join(s)
    with gzip.open(path, "wb", compresslevel=6) as f:
        for i, s in enumerate(seqs):
            q = (rng.integers(2, 41, len(s)) + 33).astype(np.uint8)
            if i % 5 == 0:
                q[0] = ord("@")
            f.write(b"@read%06d extra words\n" % i + s.encode() + b"\n+\n" + q.tobytes() + b"\n")
    return seqs


_OPEN = []


@pytest.fixture(autouse=True)
def _close_engines():
    yield
    while _OPEN:
        s = _OPEN.pop()
        if getattr(s, "_engine", None) is not None:
            s._engine.close()


def _run(fq, tmp, hmm, stream, fast, monkeypatch):
    os.makedirs(tmp, exist_ok=True)
    monkeypatch.setenv("ITSXPRESS_GPUS", "1")
    monkeypatch.setenv("ITSXPRESS_STREAM", "1" if stream else "0")
    monkeypatch.setenv("ITSXPRESS_ARRAYS", "1" if fast else "0")
    from itsxpress_amd import trim
    trim.cache_clear()
    sobj = S.SeqSampleNotPaired(fastq=fq, tempdir=tmp)
    _OPEN.append(sobj)
    sobj.deduplicate(threads=1)
    sobj._search(hmmfile=hmm, threads=1)
    its_pos = S.ItsPosition(domtable=sobj.dom_file, region="ITS2")
    dedup_obj = S.Dedup(uc_file=sobj.uc_file, rep_file=sobj.rep_file, seq_file=sobj.seq_file, fastq=sobj.r1, fastq2=sobj.fastq2)
    out = os.path.join(tmp, "trimmed.fq.gz")
    dedup_obj.create_trimmed_seqs(out, gzipped=True, zstd_file=False, itspos=its_pos, wri_file=True, tempdir=tmp)
    coords = sobj.trim_coordinates("ITS2")
    return sobj, open(out, "rb").read(), [np.asarray(c).copy() for c in coords], its_pos, dedup_obj


def test_chunks_equal_one_engine_file_for_file(tmp_path, t_hmm_text, monkeypatch):
    tmp = str(tmp_path)
    hmm = _its2(tmp, t_hmm_text)
    fq = os.path.join(tmp, "synth.fq.gz")
    _fastq_gz(fq, t_hmm_text, 12000, 515)
    # small inflater rounds and small chunks: the file arrives in ~8 slices while it is being inflated
    monkeypatch.setenv("ITSX_PINFLATE_CHUNK_KB", "32")
    monkeypatch.setenv("ITSX_STREAM_CHUNK_MB", "1.5")
    one, out1, c1, pos1, dd1 = _run(fq, os.path.join(tmp, "one"), hmm, False, False, monkeypatch)
    st, out2, c2, pos2, dd2 = _run(fq, os.path.join(tmp, "stream"), hmm, True, False, monkeypatch)
    from itsxpress_amd.stream import StreamEngine
    assert isinstance(st._engine, StreamEngine) and st._engine.world >= 4, st._engine.world
    for name in ("uc.txt", "rep.fa", "domtbl.txt"):
        a = open(os.path.join(tmp, "one", name), "rb").read()
        b = open(os.path.join(tmp, "stream", name), "rb").read()
        assert len(a) > 100 and a == b, name
    assert out1 == out2 and len(out1) > 1000
    assert all(np.array_equal(x, y) for x, y in zip(c1, c2))
    assert pos1.ddict == pos2.ddict and dd1.matchdict == dd2.matchdict
    # arrays mode: the pipeline proper (load, derep and lazy search overlapped; nothing written but the trimmed reads)
    sf, outf, cf, posf, ddf = _run(fq, os.path.join(tmp, "fast"), hmm, True, True, monkeypatch)
    eng = sf._engine
    assert isinstance(eng, StreamEngine) and eng.world >= 4
    assert outf == out1 and all(np.array_equal(x, y) for x, y in zip(c1, cf))
    assert not os.path.exists(os.path.join(tmp, "fast", "uc.txt")) and not os.path.exists(os.path.join(tmp, "fast", "domtbl.txt"))
    assert ddf.matchdict == dd1.matchdict
    # the search of every chunk ran inside the pipeline (timeline: chunk, text ready, loaded, searched) and only first occurrences
    # were scored: the chunks' scored uniques add up to the sample's
    assert len(eng.timeline) == eng.world and all(t[3] >= t[2] >= t[1] for t in eng.timeline)
    assert sum(int(((v[:, 2] == k) & (v[:, 3] == np.arange(v.shape[0]))).sum()) for k, v in enumerate(eng._verdicts)) == eng.n_unique


def test_reference_fixture_and_one_slice_files(tmp_path, t_hmm_text, monkeypatch):
    """the reference's own fixture (227 reads: one slice, one chunk) and a plain-text file through the streaming engine"""
    tmp = str(tmp_path)
    hmm = _its2(tmp, t_hmm_text)
    fqz = os.path.join(GOLD, "seq.fq.gz")
    one, out1, c1, pos1, dd1 = _run(fqz, os.path.join(tmp, "one"), hmm, False, False, monkeypatch)
    st, out2, c2, pos2, dd2 = _run(fqz, os.path.join(tmp, "stream"), hmm, True, False, monkeypatch)
    assert st._engine.world == 1
    for name in ("uc.txt", "rep.fa", "domtbl.txt"):
        assert open(os.path.join(tmp, "one", name), "rb").read() == open(os.path.join(tmp, "stream", name), "rb").read()
    assert out1 == out2 and all(np.array_equal(x, y) for x, y in zip(c1, c2))
    plain = os.path.join(tmp, "seq.fq")
    with gzip.open(fqz, "rb") as f, open(plain, "wb") as g:
        g.write(f.read())
    sp, out3, c3, _, _ = _run(plain, os.path.join(tmp, "plain"), hmm, True, True, monkeypatch)
    assert out3 == out1 and all(np.array_equal(x, y) for x, y in zip(c1, c3))
    # ITSXPRESS_STREAM unset: a large input streams by itself in arrays mode (and only there; ITSX_STREAM_AUTO_MB is the size, 512)
    from itsxpress_amd.stream import StreamEngine
    from itsxpress_amd.engine import Engine as _E
    monkeypatch.delenv("ITSXPRESS_STREAM")
    monkeypatch.setenv("ITSX_STREAM_AUTO_MB", "0.01")
    for arrays, want in (("1", StreamEngine), ("0", _E)):
        monkeypatch.setenv("ITSXPRESS_ARRAYS", arrays)
        sa = S.SeqSampleNotPaired(fastq=plain, tempdir=os.path.join(tmp, "auto" + arrays))
        _OPEN.append(sa)
        assert type(sa.engine) is want
    monkeypatch.setenv("ITSX_STREAM_AUTO_MB", "512")
    sa = S.SeqSampleNotPaired(fastq=plain, tempdir=os.path.join(tmp, "auto_small"))
    monkeypatch.setenv("ITSXPRESS_ARRAYS", "1")
    _OPEN.append(sa)
    assert type(sa.engine) is _E                              # (a 150-KB file)
    # greedy clustering (cluster_id < 1) is sequential by definition: under ITSXPRESS_STREAM the mirror runs it on one plain engine
    from itsxpress_amd.engine import Engine
    monkeypatch.setenv("ITSXPRESS_STREAM", "1")
    sc = S.SeqSampleNotPaired(fastq=plain, tempdir=os.path.join(tmp, "cl"))
    os.makedirs(sc.tempdir, exist_ok=True)
    _OPEN.append(sc)
    sc.cluster(threads=1, cluster_id=0.995)
    assert type(sc._engine) is Engine and sc._engine.n_unique > 0


def test_corrupt_gzip_is_an_error_not_a_result(tmp_path, t_hmm_text, monkeypatch):
    """a flipped byte deep in the file: slices handed out before it are void -- the run ends in an error, as the plain loader's"""
    tmp = str(tmp_path)
    hmm = _its2(tmp, t_hmm_text)
    fq = os.path.join(tmp, "synth.fq.gz")
    _fastq_gz(fq, t_hmm_text, 8000, 99)
    raw = bytearray(open(fq, "rb").read())
    raw[len(raw) * 3 // 4] ^= 0x55
    bad = os.path.join(tmp, "bad.fq.gz")
    open(bad, "wb").write(bytes(raw))
    monkeypatch.setenv("ITSX_PINFLATE_CHUNK_KB", "32")
    monkeypatch.setenv("ITSX_STREAM_CHUNK_MB", "1")
    from itsxpress_amd import EngineError
    with pytest.raises(EngineError):
        _run(bad, os.path.join(tmp, "bad"), hmm, True, True, monkeypatch)


def test_writer_inside_the_pipeline_writes_the_same_bytes(tmp_path, t_hmm_text, monkeypatch):
    """SeqSample.plan_output before deduplicate(): chunks are finalized with provisional domZ bounds right after their own search
    and their reads deflated while later chunks are scored; the reads of representatives with an undecided row wait for the exact
    thresholds.  The file is byte for byte what the writer produces in one go after everything else -- also when MANY rows are
    undecided at first (ITSX_LAZY_ZUB_SCALE widens the bounds)."""
    tmp = str(tmp_path)
    hmm = _its2(tmp, t_hmm_text)
    monkeypatch.setenv("ITSX_PINFLATE_CHUNK_KB", "32")
    monkeypatch.setenv("ITSX_STREAM_CHUNK_MB", "1.5")
    monkeypatch.setenv("ITSX_WRITE_UNIT_KB", "256")
    from itsxpress_amd.stream import StreamEngine
    for name, scale in (("plain", None), ("wide", "20000")):
        fq = os.path.join(tmp, name + ".fq.gz")
        _fastq_gz(fq, t_hmm_text, 12000, 616, damage=0.6 if scale else 0.0)
        if scale:
            monkeypatch.setenv("ITSX_LAZY_ZUB_SCALE", scale)
        _, ref_bytes, c_ref, _, _ = _run(fq, os.path.join(tmp, "one_" + name), hmm, False, True, monkeypatch)
        d = os.path.join(tmp, name)
        os.makedirs(d, exist_ok=True)
        monkeypatch.setenv("ITSXPRESS_GPUS", "1"); monkeypatch.setenv("ITSXPRESS_STREAM", "1"); monkeypatch.setenv("ITSXPRESS_ARRAYS", "1")
        from itsxpress_amd import trim
        trim.cache_clear()
        sobj = S.SeqSampleNotPaired(fastq=fq, tempdir=d)
        _OPEN.append(sobj)
        out = os.path.join(d, "trimmed.fq.gz")
        sobj.plan_output(out, "ITS2", gzipped=True)
        sobj.deduplicate(threads=1)
        sobj._search(hmmfile=hmm, threads=1)
        eng = sobj._engine
        assert isinstance(eng, StreamEngine) and eng.world >= 4 and eng._out is not None
        # the writer has been at work during the pipeline: the file exists before anybody asked for it
        assert os.path.exists(out)
        its_pos = S.ItsPosition(domtable=sobj.dom_file, region="ITS2")
        dd = S.Dedup(uc_file=sobj.uc_file, rep_file=sobj.rep_file, seq_file=sobj.seq_file, fastq=sobj.r1, fastq2=sobj.fastq2)
        dd.create_trimmed_seqs(out, gzipped=True, zstd_file=False, itspos=its_pos, wri_file=True, tempdir=d)
        assert open(out, "rb").read() == ref_bytes, name
        assert all(np.array_equal(x, y) for x, y in zip(c_ref, sobj.trim_coordinates("ITS2")))
        if scale:
            assert eng._out.n_late_uniques > 0, "the wide bounds were meant to leave rows undecided at first"
            monkeypatch.delenv("ITSX_LAZY_ZUB_SCALE")
        # another output path than the planned one: written the ordinary way
        other = os.path.join(d, "other.fq.gz")
        dd.create_trimmed_seqs(other, gzipped=True, zstd_file=False, itspos=its_pos, wri_file=True, tempdir=d)
        assert open(other, "rb").read() == ref_bytes


def test_paired_end_through_the_streaming_engine(tmp_path, t_hmm_text, monkeypatch):
    """the reference's paired fixture.  File mode: merge (one plain engine does it), then the merged file -- its text is in the cache,
    the stream cuts it from there -- dereplicated and scored in chunks.  Arrays mode (round 6): R1 and R2 streamed side by side, R2 cut
    at R1's record counts, every pair of slices merged by its chunk's context; with SeqSample.plan_output_paired the two outputs are
    written by the pipeline itself.  R1 / R2 trimmed with the merged reads' coordinates: same files as without ITSXPRESS_STREAM."""
    tmp = str(tmp_path)
    hmm = _its2(tmp, t_hmm_text)
    raw = []
    for fn in ("4774-1-MSITS3_R1.fastq", "4774-1-MSITS3_R2.fastq"):
        p = os.path.join(tmp, fn)
        with gzip.open(os.path.join(GOLD, fn + ".gz"), "rb") as f, open(p, "wb") as g:
            g.write(f.read())
        raw.append(p)
    monkeypatch.setenv("ITSX_STREAM_CHUNK_MB", "0.02")           # the merged file is ~100 KB: a handful of chunks
    outs = {}
    for name, stream, arrays in (("one", False, "1"), ("stream", True, "0"), ("stream_arrays", True, "1"), ("stream_arrays_planned", True, "1")):
        d = os.path.join(tmp, name)
        os.makedirs(d, exist_ok=True)
        monkeypatch.setenv("ITSXPRESS_GPUS", "1")
        monkeypatch.setenv("ITSXPRESS_STREAM", "1" if stream else "0")
        monkeypatch.setenv("ITSXPRESS_ARRAYS", arrays)
        sobj = S.SeqSamplePairedNotInterleaved(fastq=raw[0], tempdir=d, fastq2=raw[1])
        _OPEN.append(sobj)
        o1, o2 = os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")
        if name.endswith("planned"):
            sobj.plan_output_paired(o1, o2, "ITS2")
        sobj._merge_reads(threads=1)
        sobj.deduplicate(threads=1)
        sobj._search(hmmfile=hmm, threads=1)
        its_pos = S.ItsPosition(domtable=sobj.dom_file, region="ITS2")
        dd = S.Dedup(uc_file=sobj.uc_file, rep_file=sobj.rep_file, seq_file=sobj.seq_file, fastq=sobj.r1, fastq2=sobj.fastq2)
        dd.create_paired_trimmed_seqs(o1, o2, gzipped=False, zstd_file=False, itspos=its_pos, wri_file=True)
        outs[name] = (open(o1, "rb").read(), open(o2, "rb").read(), [np.asarray(c).copy() for c in sobj.trim_coordinates("ITS2")])
        if name == "one":
            merged_one = sobj._engine.n_reads
        from itsxpress_amd.stream import StreamEngine
        from itsxpress_amd.engine import Engine as _E
        if stream and arrays == "0":                     # through seq.fq: the merged file is cut from the cache, chunk by chunk
            assert isinstance(sobj._engine, StreamEngine) and sobj._engine.world >= 3, sobj._engine.world
        if stream and arrays == "1":                     # arrays mode: R1 / R2 streamed, merged chunk by chunk (no seq.fq at all)
            assert isinstance(sobj._engine, StreamEngine) and sobj._engine.world >= 3 and not os.path.exists(os.path.join(d, "seq.fq"))
            assert sobj._engine.n_pairs == 250 and sobj._engine.n_reads == merged_one     # (the fixture's 250 pairs; as many merged as in one context)
            if name.endswith("planned"):
                assert sobj._engine._out is not None and sobj._engine._out.result is not None
    for other in ("stream", "stream_arrays", "stream_arrays_planned"):
        assert outs["one"][0] == outs[other][0] and outs["one"][1] == outs[other][1] and len(outs["one"][0]) > 1000
        assert all(np.array_equal(x, y) for x, y in zip(outs["one"][2], outs[other][2]))


def test_synthetic_pairs_streamed_equal_the_staged_run(tmp_path, t_hmm_text, monkeypatch):
    """30 000 synthetic 2x250 pairs (scripts/paired_run.py's generator: sequencing errors, pairs that do not merge), .fastq.gz in and out,
    several chunks: the streamed pipeline's two outputs, inflated, equal the staged run's byte for byte; R2 shorter than R1 is an error"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import paired_run
    monkeypatch.setenv("ITSX_STREAM_CHUNK_MB", "2")
    monkeypatch.setenv("ITSX_PINFLATE_CHUNK_KB", "256")
    res = paired_run.run(30000, True, stream=True, check=True)
    st = res["streamed"]
    assert st["outputs_equal_staged"] and st["chunks"] >= 3 and st["pairs"] == 30000 and st["merged"] == res["merged"]
    assert res["pairs_written"] > 20000


def test_stream_that_outgrows_its_reservation_starts_over_serially(tmp_path, t_hmm_text, monkeypatch):
    """the block-parallel inflater gives up after slices are out (the reservation is made too small on purpose): the engine drops
    what it computed -- the planned output included -- and runs the file again through the serial inflater; same bytes"""
    tmp = str(tmp_path)
    hmm = _its2(tmp, t_hmm_text)
    fq = os.path.join(tmp, "synth.fq.gz")
    _fastq_gz(fq, t_hmm_text, 12000, 717)
    monkeypatch.setenv("ITSX_PINFLATE_CHUNK_KB", "32")
    monkeypatch.setenv("ITSX_STREAM_CHUNK_MB", "1.5")
    _, ref_bytes, c_ref, _, _ = _run(fq, os.path.join(tmp, "one"), hmm, False, True, monkeypatch)
    monkeypatch.setenv("ITSX_STREAM_RESERVE_X", "1")
    monkeypatch.setenv("ITSX_STREAM_RESERVE_MB", "4")
    monkeypatch.setenv("ITSXPRESS_GPUS", "1"); monkeypatch.setenv("ITSXPRESS_STREAM", "1"); monkeypatch.setenv("ITSXPRESS_ARRAYS", "1")
    from itsxpress_amd import trim, _lib
    trim.cache_clear()
    d = os.path.join(tmp, "stream")
    os.makedirs(d, exist_ok=True)
    before = _lib.lib().itsx_io_parallel_inflates()
    sobj = S.SeqSampleNotPaired(fastq=fq, tempdir=d)
    _OPEN.append(sobj)
    out = os.path.join(d, "trimmed.fq.gz")
    sobj.plan_output(out, "ITS2", gzipped=True)
    sobj.deduplicate(threads=1)
    sobj._search(hmmfile=hmm, threads=1)
    assert _lib.lib().itsx_io_parallel_inflates() == before          # the parallel attempt did not deliver the file
    assert os.environ.get("ITSX_PARALLEL_INFLATE") is None           # (the driver's switch for its second attempt is gone again)
    its_pos = S.ItsPosition(domtable=sobj.dom_file, region="ITS2")
    dd = S.Dedup(uc_file=sobj.uc_file, rep_file=sobj.rep_file, seq_file=sobj.seq_file, fastq=sobj.r1, fastq2=sobj.fastq2)
    dd.create_trimmed_seqs(out, gzipped=True, zstd_file=False, itspos=its_pos, wri_file=True, tempdir=d)
    assert open(out, "rb").read() == ref_bytes
    assert all(np.array_equal(x, y) for x, y in zip(c_ref, sobj.trim_coordinates("ITS2")))
