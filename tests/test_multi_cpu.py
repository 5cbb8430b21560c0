"""ITSXPRESS_GPUS=N (itsxpress_amd/multi.py) without a GPU: the owner's step in numpy against dist.py's torch statement, the array
writers (csrc/writers_host.cpp) against the reference's frozen uc.txt / rep.fa, and the error path of the worker pool."""
import gzip
import os

import numpy as np
import pytest

from conftest import GOLD


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGTUacgtu", "TGCAAtgcaa"))


def test_numpy_owner_step_equals_the_torch_one():
    import torch
    from itsxpress_amd import dist, multi
    rng = np.random.default_rng(5)
    for world in (2, 4, 7):
        m = 4000
        keys = rng.integers(-2 ** 62, 2 ** 62, (300, 2))              # 300 distinct sequences, many holders each
        pick = rng.integers(0, 300, m)
        gidx = rng.permutation(10 * m)[:m].astype(np.int64)
        recv = np.stack([keys[pick, 0], keys[pick, 1], gidx, rng.integers(0, 2, m), rng.integers(0, 10 ** 6, m)], axis=1).astype(np.int64)
        src = rng.integers(0, world, m).astype(np.int64)
        a = multi.owner_verdicts(recv, src)
        b = dist.owner_verdicts(torch.from_numpy(recv), torch.from_numpy(src)).numpy()
        assert np.array_equal(a, multi.owner_verdicts_native(recv, src))          # (the library's: what the workers run)
        assert np.array_equal(a, b)
        # one scorer per distinct sequence, in the representative's orientation, and the representative is the smallest index
        for k in range(300):
            rows = np.nonzero(pick == k)[0]
            if len(rows) == 0:
                continue
            assert len(set(map(tuple, a[rows]))) == 1
            first = rows[np.argmin(gidx[rows])]
            assert a[rows[0], 0] == gidx[first] and a[rows[0], 1] == recv[first, 3]
            holders = [(int(src[r]), int(recv[r, 4])) for r in rows if recv[r, 3] == recv[first, 3]]
            assert (int(a[rows[0], 2]), int(a[rows[0], 3])) in holders
    assert multi.owner_verdicts(np.zeros((0, 5), np.int64), np.zeros(0, np.int64)).shape == (0, 4)
    assert multi.owner_verdicts_native(np.zeros((0, 5), np.int64), np.zeros(0, np.int64)).shape == (0, 4)
    # at a size where the library's pool is at work (buckets filled by several threads), keys of both signs, heavy duplication
    m = 200000
    keys = rng.integers(-2 ** 63, 2 ** 63 - 1, (40000, 2))
    pick = rng.integers(0, 40000, m)
    recv = np.stack([keys[pick, 0], keys[pick, 1], rng.permutation(3 * m)[:m], rng.integers(0, 2, m), rng.integers(0, 10 ** 6, m)], axis=1).astype(np.int64)
    src = rng.integers(0, 8, m).astype(np.int64)
    assert np.array_equal(multi.owner_verdicts(recv, src), multi.owner_verdicts_native(recv, src))


def test_array_writers_reproduce_the_reference_fixture(tmp_path):
    """vsearch --fastx_uniques's uc.txt / rep.fa (tests/test_data/ex_tmpdir) from per-read arrays: byte for byte"""
    import ctypes as C
    from itsxpress_amd import _lib
    from itsxpress_amd.engine import read_fastx
    names, seqs = read_fastx(os.path.join(GOLD, "seq.fq.gz"))
    first, rep_of, strand = {}, [], []
    for i, s in enumerate(seqs):
        u = s.upper()
        if u in first:
            rep_of.append(first[u]); strand.append(1)
        elif _rc(u) in first:
            rep_of.append(first[_rc(u)]); strand.append(-1)
        else:
            first[u] = i; rep_of.append(i); strand.append(1)
    rep_of = np.array(rep_of, np.int64)
    strand = np.array(strand, np.int8)
    lens = np.array([len(s) for s in seqs], np.int32)
    nb = "".join(names).encode()
    no = np.zeros(len(names) + 1, np.int64)
    np.cumsum([len(x) for x in names], out=no[1:])
    seeds = [i for i in range(len(seqs)) if rep_of[i] == i]
    sb = "".join(seqs[i] for i in seeds).encode()
    so = np.zeros(len(seeds) + 1, np.int64)
    np.cumsum([len(seqs[i]) for i in seeds], out=so[1:])
    uc, rep = str(tmp_path / "uc.txt"), str(tmp_path / "rep.fa")
    L = _lib.lib()
    rc = L.itsx_write_derep_arrays(uc.encode(), rep.encode(), len(seqs), rep_of.ctypes.data, strand.ctypes.data, lens.ctypes.data,
                                   nb, no.ctypes.data, sb, so.ctypes.data, len(seeds))
    assert rc == 0, L.itsx_writers_last_error()
    assert open(uc, "rb").read() == open(os.path.join(GOLD, "fixture_uc.txt"), "rb").read()
    assert open(rep, "rb").read() == open(os.path.join(GOLD, "fixture_rep.fa"), "rb").read()
    # a seed count that does not match the sequences given is refused, not written
    assert L.itsx_write_derep_arrays(None, rep.encode(), len(seqs), rep_of.ctypes.data, strand.ctypes.data, lens.ctypes.data,
                                     nb, no.ctypes.data, sb, so.ctypes.data, len(seeds) - 1) < 0
    assert b"seeds" in L.itsx_writers_last_error()


def test_worker_pool_fails_loudly_without_a_gpu():
    """no CPU fallback: without a gfx950 device the workers cannot create their contexts -- the driver reports it and leaves no
    process behind (here: no GPU at all; on a GPU box the same path is what a missing device id takes)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from itsxpress_amd import EngineError
    from itsxpress_amd.multi import MultiEngine
    with pytest.raises(EngineError) as e:
        MultiEngine(2)
    assert "no CPU fallback" in str(e.value) or "device" in str(e.value).lower()


def test_mirror_reads_its_switches_from_the_environment(monkeypatch):
    import importlib
    S = importlib.import_module("itsxpress_amd.SeqSample")
    from itsxpress_amd.multi import gpus_from_env
    monkeypatch.delenv("ITSXPRESS_GPUS", raising=False)
    monkeypatch.delenv("ITSXPRESS_ARRAYS", raising=False)
    assert gpus_from_env() == 1
    s = S.SeqSampleNotPaired("x.fq", "/tmp")
    assert not s._is_fast()
    monkeypatch.setenv("ITSXPRESS_ARRAYS", "1")
    monkeypatch.setenv("ITSXPRESS_GPUS", "8")
    assert s._is_fast() and gpus_from_env() == 8
    s.fast = False
    assert not s._is_fast()
    t = S.EngineTable("/tmp/uc.txt", object(), "uc")
    assert t == "/tmp/uc.txt" and os.path.basename(t) == "uc.txt" and t.kind == "uc"


def test_install_runs_file_compatible_whatever_the_environment_says(tmp_path, monkeypatch):
    """install() leaves the reference's own ItsPosition / Dedup in place; they open uc.txt / domtbl.txt by path, so the installed
    methods must write those files even with ITSXPRESS_ARRAYS=1 in the environment (advisor, round 4)"""
    import sys
    import types
    pkg, mod = types.ModuleType("itsxpress"), types.ModuleType("itsxpress.SeqSample")

    class RefSample:
        def __init__(self, fastq, tempdir):
            self.fastq, self.tempdir, self.seq_file = fastq, tempdir, fastq

    class RefPaired(RefSample):
        pass
    mod.SeqSample, mod.SeqSamplePairedNotInterleaved = RefSample, RefPaired
    pkg.SeqSample = mod
    monkeypatch.setitem(sys.modules, "itsxpress", pkg)
    monkeypatch.setitem(sys.modules, "itsxpress.SeqSample", mod)
    monkeypatch.setenv("ITSXPRESS_ARRAYS", "1")
    from itsxpress_amd.SeqSample import install, SeqSample
    install()
    s = RefSample("x.fq", str(tmp_path))
    assert s._is_fast() is False
    assert SeqSample("x.fq", str(tmp_path))._is_fast() is True            # the mirror itself still follows the switch


def test_shard_text_cuts_at_record_starts_and_follows_the_mate_file(tmp_path):
    """itsx_shard_text (csrc/shard_host.cpp): the pieces are whole records, in order, and together the file; a mate file is cut at the
    same record counts whatever its bytes; qualities that look like titles do not fool the cut; gzip input; FASTA; more pieces than
    records; a mate that does not hold the records is refused"""
    import ctypes as C
    import gzip
    from itsxpress_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)

    def fastq(n, lo, hi, tag):
        recs = []
        for i in range(n):
            ln = int(rng.integers(lo, hi))
            seq = "".join(rng.choice(list("ACGT"), ln))
            qual = "".join(rng.choice(list("@+I5"), ln))                  # '@' and '+' at line starts on purpose
            recs.append("@%s%d some text\n%s\n+\n%s\n" % (tag, i, seq, qual))
        return recs

    def shard(path, parts, match=None):
        rec = np.zeros(parts, np.int64)
        by = np.zeros(parts, np.int64)
        pre = str(tmp_path / ("piece_%d" % rng.integers(1 << 30)))
        m = None if match is None else np.ascontiguousarray(match, np.int64)
        rc = L.itsx_shard_text(os.fsencode(path), parts, None if m is None else m.ctypes.data, os.fsencode(pre), rec.ctypes.data, by.ctypes.data)
        return rc, rec, [open("%s.%d" % (pre, p), "rb").read() if rc == 0 else b"" for p in range(parts)]

    r1 = fastq(1000, 30, 400, "a")
    r2 = fastq(1000, 200, 210, "b")
    p1, p2 = tmp_path / "r1.fq", tmp_path / "r2.fq.gz"
    p1.write_text("".join(r1))
    with gzip.open(p2, "wt") as f:
        f.write("".join(r2))
    for parts in (1, 2, 3, 8):
        rc, rec, pieces = shard(str(p1), parts)
        assert rc == 0 and rec.sum() == 1000 and b"".join(pieces) == "".join(r1).encode()
        at = 0
        for p in range(parts):
            assert pieces[p] == "".join(r1[at:at + rec[p]]).encode()
            at += rec[p]
        rc, rec2, pieces2 = shard(str(p2), parts, match=rec)
        assert rc == 0 and np.array_equal(rec, rec2) and b"".join(pieces2) == "".join(r2).encode()
        at = 0
        for p in range(parts):
            assert pieces2[p] == "".join(r2[at:at + rec[p]]).encode()
            at += rec[p]
    # more pieces than records; no trailing newline
    small = tmp_path / "small.fq"
    small.write_text("".join(r1[:3]).rstrip("\n"))
    rc, rec, pieces = shard(str(small), 7)
    assert rc == 0 and rec.sum() == 3 and b"".join(pieces) == "".join(r1[:3]).rstrip("\n").encode()
    # FASTA with wrapped sequences
    fa = tmp_path / "x.fa"
    recs = [">s%d\n%s\n%s\n" % (i, "ACGT" * 15, "GG" * int(rng.integers(1, 20))) for i in range(200)]
    fa.write_text("".join(recs))
    rc, rec, pieces = shard(str(fa), 4)
    assert rc == 0 and rec.sum() == 200 and b"".join(pieces) == "".join(recs).encode() and all(pc[:1] in (b">", b"") for pc in pieces)
    rc, rec2, pieces2 = shard(str(fa), 4, match=rec)
    assert rc == 0 and np.array_equal(rec, rec2) and pieces2 == pieces
    # a mate with fewer records
    short = tmp_path / "short.fq"
    short.write_text("".join(r2[:900]))
    rc, _, _ = shard(str(short), 3, match=np.array([400, 300, 300]))
    assert rc != 0 and b"mate" in L.itsx_shard_last_error()
