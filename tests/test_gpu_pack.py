"""The hand-over of reads (itsx_set_reads / _view / _device): chunked staging, the exception-list resize and the
device-resident entry point must all leave the same packed reads in HBM."""
import os

import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu


def _reads():
    rng = np.random.default_rng(9)
    seqs = []
    for i in range(3000):
        L = int(rng.integers(1, 700)) if i % 50 else int(rng.integers(0, 3))
        s = rng.choice(list("ACGT"), L)
        if i % 7 == 0 and L:                                # IUPAC-rich reads: exceptions in many words
            k = rng.integers(0, L, max(1, L // 5))
            s[k] = rng.choice(list("NRYKMSWBDHVnacgtu"), k.size)
        seqs.append("".join(s))
    seqs.append("N" * 400)
    seqs.append("")
    return seqs


def _snapshot(eng, n):
    hf, hr = eng.debug_read_hashes()
    packed = [eng.debug_packed_read(i) for i in range(0, n, 37)]
    return hf.copy(), hr.copy(), packed


def _same(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for (w1, e1), (w2, e2) in zip(a[2], b[2]):
        assert np.array_equal(w1, w2) and np.array_equal(e1, e2)


def test_chunked_staging_exception_resize_and_device_text_agree(engine, monkeypatch):
    import torch
    seqs = _reads()
    n = len(seqs)
    lens = np.array([len(s) for s in seqs], np.int64)
    offs = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    blob = "".join(seqs).encode()
    engine.set_reads_buffer(blob, offs)
    ref = _snapshot(engine, n)
    # the packed words equal the oracle's digitisation, read by read
    codes, o = orc.digitize(seqs)
    for i, (w, e) in zip(range(0, n, 37), ref[2]):
        c = np.asarray(codes[o[i]:o[i] + lens[i]])
        exp_w = np.zeros(max(1, (lens[i] + 15) // 16), np.uint32)
        for p, x in enumerate(c):
            if x <= 3:
                exp_w[p >> 4] |= np.uint32(int(x) << (2 * (p & 15)))
        assert np.array_equal(w, exp_w)
        assert np.array_equal(e, np.array([(p << 4) | int(x) for p, x in enumerate(c) if x > 3], np.uint32))
    monkeypatch.setenv("ITSX_PACK_CHUNK", "70000")          # dozens of chunks through the three staging buffers
    engine.set_reads_buffer(blob, offs)
    _same(ref, _snapshot(engine, n))
    monkeypatch.setenv("ITSX_PACK_ECAP", "1")               # the exception list is too small: sized exactly, pass repeated
    engine.set_reads_buffer(np.frombuffer(blob, np.uint8), offs)
    _same(ref, _snapshot(engine, n))
    monkeypatch.delenv("ITSX_PACK_CHUNK")
    engine.set_reads_buffer(blob, offs)
    _same(ref, _snapshot(engine, n))
    monkeypatch.delenv("ITSX_PACK_ECAP")
    # text already in device memory
    d = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    engine.set_reads_device(d.data_ptr(), offs, keep=d)
    _same(ref, _snapshot(engine, n))


def test_device_text_gives_the_same_files(engine, t_hmm_text, tmp_path):
    import torch
    blob, offs = synth.make_reads(t_hmm_text, 2000, seed=41, fixed_len=0, len_range=(200, 420))
    outs = []
    for mode in ("host", "device"):
        if mode == "host":
            engine.set_reads_buffer(blob, offs)
        else:
            d = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
            torch.cuda.synchronize()
            engine.set_reads_device(d.data_ptr(), offs, keep=d)
        engine.derep()
        uc, rep = str(tmp_path / (mode + ".uc")), str(tmp_path / (mode + ".fa"))
        engine.write_uc(uc)
        engine.write_rep_fasta(rep)
        outs.append((open(uc).read(), open(rep).read()))
    assert outs[0] == outs[1] and len(outs[0][1]) > 1000


def test_illegal_symbol_is_reported_from_any_chunk(engine, monkeypatch):
    from itsxpress_amd import EngineError
    seqs = ["ACGT" * 100] * 500
    seqs[417] = "ACGT" * 50 + "!" + "ACGT" * 49
    offs = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=offs[1:])
    monkeypatch.setenv("ITSX_PACK_CHUNK", "70000")
    with pytest.raises(EngineError, match="read 417"):
        engine.set_reads_buffer("".join(seqs).encode(), offs)
