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
