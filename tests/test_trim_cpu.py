"""f1: the native FASTQ slice/write path against the reference's byte-compared goldens
(tests/test_main_pytest.py:68-161, 350-397).  No GPU needed: the writers are host-only and context-free."""
import gzip
import os

import numpy as np
import pytest

from itsxpress_amd import Dedup, ItsPosition, EngineError
from itsxpress_amd.engine import read_fastx
from itsxpress_amd.trim import write_trimmed_fastq, write_trimmed_paired


def _read(path):
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rt") as f:
        return f.read()


@pytest.fixture(scope="module")
def golden_coords(gold):
    rows = [ln.split("\t") for ln in open(os.path.join(gold, "fungi_its2_coords.tsv")).read().strip().split("\n")[1:]]
    return {r[0]: (int(r[2]), int(r[3]), int(r[4])) for r in rows}


@pytest.fixture(scope="module")
def raw_fastqs(gold, tmp_path_factory):
    d = tmp_path_factory.mktemp("raw")
    out = []
    for fn in ("4774-1-MSITS3_R1.fastq", "4774-1-MSITS3_R2.fastq"):
        p = d / fn
        p.write_text(_read(os.path.join(gold, fn + ".gz")))
        out.append(str(p))
    return out


def test_paired_writer_reproduces_reference_goldens_byte_for_byte(gold, golden_coords, raw_fastqs, tmp_path):
    names, _ = read_fastx(os.path.join(gold, "seq.fq.gz"))           # the 227 merged reads
    start = np.array([golden_coords.get(n, (-1, -1, -1))[0] for n in names], np.int32)
    stop = np.array([golden_coords.get(n, (-1, -1, -1))[1] for n in names], np.int32)
    tlen = np.array([golden_coords.get(n, (-1, -1, -1))[2] for n in names], np.int32)
    o1, o2 = str(tmp_path / "t2_r1.fq"), str(tmp_path / "t2_r2.fq")
    n = write_trimmed_paired(raw_fastqs[0], raw_fastqs[1], o1, o2, names, start, stop, tlen)
    assert n == 226
    assert open(o1).read() == _read(os.path.join(gold, "t2_r1.fq.gz"))
    assert open(o2).read() == _read(os.path.join(gold, "t2_r2.fq.gz"))
    # gz inputs and gz outputs carry the same bytes
    g1, g2 = str(tmp_path / "a.fq.gz"), str(tmp_path / "b.fq.gz")
    n = write_trimmed_paired(os.path.join(gold, "4774-1-MSITS3_R1.fastq.gz"), os.path.join(gold, "4774-1-MSITS3_R2.fastq.gz"),
                             g1, g2, names, start, stop, tlen, gzipped=True)
    assert n == 226 and _read(g1) == open(o1).read() and _read(g2) == open(o2).read()


def test_single_end_writer_matches_reference_counts(gold, golden_coords, tmp_path):
    """test_dedup_create_trimmed_seqs asserts n == 226 and a summed length of 42637 on seq.fq.gz."""
    seq = os.path.join(gold, "seq.fq.gz")
    names, seqs = read_fastx(seq)
    md = Dedup(os.path.join(gold, "fixture_uc.txt"), "", seq).matchdict
    start = np.array([golden_coords.get(md[n], (-1, -1, -1))[0] for n in names], np.int32)
    stop = np.array([golden_coords.get(md[n], (-1, -1, -1))[1] for n in names], np.int32)
    out = str(tmp_path / "trimmed.fastq")
    n, tot = write_trimmed_fastq(seq, out, start, stop)
    assert (n, tot) == (226, 42637)
    recs = open(out).read().split("\n")
    assert len(recs) == 226 * 4 + 1 and all(recs[i] == "+" for i in range(2, 226 * 4, 4))
    k = names.index(recs[0][1:].split()[0])
    assert recs[1] == seqs[k][start[k]:stop[k]]
    outz = str(tmp_path / "trimmed.fastq.gz")
    assert write_trimmed_fastq(seq, outz, start, stop, gzipped=True) == (226, 42637)
    assert _read(outz) == open(out).read()


def test_mirror_dedup_writers_take_the_reference_dict_route(gold, golden_coords, raw_fastqs, tmp_path):
    """Dedup.create_paired_trimmed_seqs / create_trimmed_seqs with an ItsPosition whose ddict holds the golden rows."""
    seq = os.path.join(gold, "seq.fq.gz")
    dd = Dedup(os.path.join(gold, "fixture_uc.txt"), os.path.join(gold, "fixture_rep.fa"), seq, fastq=raw_fastqs[0], fastq2=raw_fastqs[1])
    ip = ItsPosition(None, "ITS2")
    for rid, (a, b, t) in golden_coords.items():
        rep = dd.matchdict[rid]
        ip.ddict[rep] = {"left": {"score": 50.0, "to_pos": a, "from_pos": a - 44},
                         "right": {"score": 50.0, "to_pos": b + 45, "from_pos": b + 1}, "tlen": t}
    o1, o2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    dd.create_paired_trimmed_seqs(o1, o2, False, False, ip, True)
    assert open(o1).read() == _read(os.path.join(gold, "t2_r1.fq.gz"))
    assert open(o2).read() == _read(os.path.join(gold, "t2_r2.fq.gz"))
    o3 = str(tmp_path / "single.fq")
    dd.create_trimmed_seqs(o3, False, False, ip, True, str(tmp_path))
    assert open(o3).read().count("\n+\n") == 226
    o4 = str(tmp_path / "single.fq.zst")              # zstd_file=True, as the reference's .zst outputs
    dd.create_trimmed_seqs(o4, False, True, ip, True, str(tmp_path))
    from itsxpress_amd.trim import read_text
    assert open(o4, "rb").read(4) == b"\x28\xb5\x2f\xfd" and read_text(o4).decode() == open(o3).read()


def test_slice_rules_match_python_semantics(tmp_path):
    """start >= stop dropped; stop beyond the read clamps; stop > tlen opens R1's slice; negative r2start wraps."""
    rng = np.random.default_rng(0)
    n = 60
    seqs = ["".join(rng.choice(list("ACGT"), int(L))) for L in rng.integers(20, 60, n)]
    quals = ["".join(chr(33 + int(q)) for q in rng.integers(2, 41, len(s))) for s in seqs]
    names = ["r%d" % i for i in range(n)]
    f1, f2 = tmp_path / "a.fastq", tmp_path / "b.fastq"
    with open(f1, "w") as a, open(f2, "w") as b:
        for nm, s, q in zip(names, seqs, quals):
            a.write("@%s 1:N:0\n%s\n+\n%s\n" % (nm, s, q))
            b.write("@%s 2:N:0\n%s\n+\n%s\n" % (nm, s[::-1], q[::-1]))
    start = rng.integers(-1, 40, n).astype(np.int32)
    stop = rng.integers(-1, 90, n).astype(np.int32)
    tlen = rng.integers(30, 80, n).astype(np.int32)
    out = tmp_path / "o.fastq"
    nw, tot = write_trimmed_fastq(str(f1), str(out), start, stop, trim_ccs=True)
    exp = []
    for i in range(n):
        if start[i] >= 0 and stop[i] >= 0 and start[i] < stop[i]:
            s, q = seqs[i][start[i]:stop[i]], quals[i][start[i]:stop[i]]
            exp.append("@%s 1:N:0\nGACAGGTACAAGAAGGA%sTTAACCCAGTCTCCAGT\n+\n%s%s%s\n" % (names[i], s, "~" * 17, q, "~" * 17))
    assert out.read_text() == "".join(exp) and nw == len(exp)
    o1, o2 = tmp_path / "p1.fastq", tmp_path / "p2.fastq"
    write_trimmed_paired(str(f1), str(f2), str(o1), str(o2), names[::-1], start[::-1], stop[::-1], tlen[::-1])
    e1, e2 = [], []
    for i in range(n):
        a, b, t = int(start[i]), int(stop[i]), int(tlen[i])
        if a >= 0 and b >= 0 and a < b:
            r2s, r2e = t - b, t - a
            s1 = slice(a, None) if b > t else slice(a, b)
            s2 = slice(r2s, None) if r2e > t else slice(r2s, r2e)
            e1.append("@%s 1:N:0\n%s\n+\n%s\n" % (names[i], seqs[i][s1], quals[i][s1]))
            e2.append("@%s 2:N:0\n%s\n+\n%s\n" % (names[i], seqs[i][::-1][s2], quals[i][::-1][s2]))
    assert o1.read_text() == "".join(e1) and o2.read_text() == "".join(e2)


def test_writer_errors(tmp_path):
    bad = tmp_path / "bad.fastq"
    bad.write_text("@r1\nACGT\n+\nII\n")
    with pytest.raises(EngineError):
        write_trimmed_fastq(str(bad), str(tmp_path / "o.fq"), [0], [3])
    with pytest.raises(FileNotFoundError):
        write_trimmed_fastq(str(tmp_path / "nope.fq"), str(tmp_path / "o.fq"), [0], [3])


def test_record_names_come_from_the_writers_own_parser(tmp_path):
    """blank lines between records: the Python line reader used to stop there while the native writer went on; names and
    records now come from one parser (itsx_fastq_ids)"""
    from itsxpress_amd.trim import read_names, write_trimmed_fastq
    p = tmp_path / "x.fastq"
    p.write_text("@a 1\nACGTACGT\n+\nIIIIIIII\n\n@b\tx\nGGGGCCCC\n+\nIIIIIIII\n\n\n@c\nTTTTAAAA\n+anything\nIIIIIIII\n")
    assert read_names(str(p)) == ["a", "b", "c"]
    out = tmp_path / "o.fastq"
    import numpy as np
    n, tot = write_trimmed_fastq(str(p), str(out), np.array([1, 0, 2], np.int32), np.array([5, 4, 8], np.int32))
    assert (n, tot) == (3, 14) and out.read_text().split("\n")[1::4][:3] == ["CGTA", "GGGG", "TTAAAA"]
    bad = tmp_path / "bad.fastq"
    bad.write_text("@a\nACGT\n+\nII\n")
    import pytest
    from itsxpress_amd import EngineError
    with pytest.raises(EngineError):
        read_names(str(bad))


def test_paired_writer_keeps_the_references_suffix_rule(tmp_path):
    import pytest
    from itsxpress_amd.trim import write_trimmed_paired
    import numpy as np
    a, b = tmp_path / "r1.fastq.gz", tmp_path / "r2.fastq"
    a.write_bytes(b""); b.write_text("")
    z = np.zeros(0, np.int32)
    with pytest.raises(ValueError, match="zstd compressed"):
        write_trimmed_paired(str(a), str(b), str(tmp_path / "o1"), str(tmp_path / "o2"), [], z, z, z)


def test_parallel_record_walk_writes_the_same_bytes(tmp_path, monkeypatch):
    """the writer object cuts the text into units at record starts (quality lines that start with '@' included) that a pool of
    threads slices and deflates; the output -- plain, with and without CCS stitching -- is byte for byte the serial walk's, the
    gzip file its content in one member per unit"""
    import numpy as np
    from itsxpress_amd.trim import write_trimmed_fastq
    rng = np.random.default_rng(3)
    fq = tmp_path / "in.fq"
    n = 6000
    with open(fq, "w") as f:
        for i in range(n):
            L = int(rng.integers(40, 400))
            seq = "".join(rng.choice(list("ACGTN"), L))
            qual = "".join(rng.choice(list("@+IF#5"), L))          # qualities that look like title and separator lines
            f.write("@read%d some words\n%s\n+\n%s\n" % (i, seq, qual))
    start = rng.integers(-1, 120, n).astype(np.int32)
    stop = rng.integers(-1, 500, n).astype(np.int32)
    out = {}
    for mode, env in (("serial", {"ITSX_WRITE_MIN_MB": "100000"}), ("parallel", {"ITSX_WRITE_MIN_MB": "0", "ITSX_WRITE_UNIT_KB": "16", "ITSX_IO_THREADS": "5"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for kind, kw in (("plain", {}), ("gz", {"gzipped": True}), ("ccs", {"trim_ccs": True})):
            o = tmp_path / ("%s_%s.out" % (mode, kind))
            out[(mode, kind)] = (write_trimmed_fastq(str(fq), str(o), start, stop, **kw), open(o, "rb").read())
        for k in env:
            monkeypatch.delenv(k)
    for kind in ("plain", "ccs"):
        assert out[("serial", kind)] == out[("parallel", kind)], kind
    assert out[("serial", "gz")][0] == out[("parallel", "gz")][0]
    assert gzip.decompress(out[("serial", "gz")][1]) == gzip.decompress(out[("parallel", "gz")][1]) == out[("serial", "plain")][1]
    assert out[("parallel", "gz")][1].count(b"\x1f\x8b\x08") > 20 >= out[("serial", "gz")][1].count(b"\x1f\x8b\x08")
    assert out[("serial", "plain")][0][0] > 1000
    # a malformed record is still named by its number
    bad = tmp_path / "bad.fq"
    with open(bad, "w") as f:
        f.write(open(fq).read())
        f.write("@broken\nACGT\n+\nII\n")
    monkeypatch.setenv("ITSX_WRITE_MIN_MB", "0")
    monkeypatch.setenv("ITSX_WRITE_UNIT_KB", "16")
    from itsxpress_amd import EngineError
    with pytest.raises(EngineError) as e:
        write_trimmed_fastq(str(bad), str(tmp_path / "x.out"), np.zeros(n + 1, np.int32), np.full(n + 1, 10, np.int32))
    assert "malformed FASTQ record %d" % n in str(e.value)


def test_writer_object_fed_piece_by_piece_writes_the_same_file(tmp_path, monkeypatch):
    """itsx_twriter_*: text and coordinates in pieces, in any rhythm, with records whose coordinates are decided late (they hold
    back their own unit only): the file is byte for byte what itsx_write_trimmed_fastq writes in one go"""
    import ctypes as C
    from itsxpress_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(8)
    n = 9000
    recs = []
    for i in range(n):
        ln = int(rng.integers(30, 380))
        q = "".join(rng.choice(list("@+IF#5"), ln))
        recs.append("@r%d w\n%s\n+\n%s\n" % (i, "".join(rng.choice(list("ACGT"), ln)), q))
    text = "".join(recs).encode()
    fq = tmp_path / "in.fq"
    fq.write_bytes(text)
    start = rng.integers(-1, 100, n).astype(np.int32)
    stop = rng.integers(-1, 420, n).astype(np.int32)
    monkeypatch.setenv("ITSX_WRITE_UNIT_KB", "64")
    monkeypatch.setenv("ITSX_IO_THREADS", "4")
    ends = np.cumsum([len(r) for r in recs])
    for kind, comp in (("gz", 1), ("plain", 0), ("zst", 2)):
        if comp == 2 and not (L.itsx_io_codecs() & 2):
            continue
        ref = tmp_path / ("ref." + kind)
        want = write_trimmed_fastq(str(fq), str(ref), start, stop, gzipped=comp == 1, zstd_file=comp == 2)
        for trial in range(3):
            out = tmp_path / ("piecewise%d.%s" % (trial, kind))
            w = C.c_void_p()
            assert L.itsx_twriter_open(os.fsencode(str(out)), comp, 0, C.byref(w)) == 0
            buf = C.create_string_buffer(text, len(text))
            base = C.addressof(buf)
            late = rng.random(n) < (0.0, 0.002, 0.05)[trial]                    # undecided when their piece arrives
            wrong = np.where(late, 7, start).astype(np.int32)                   # (whatever stands there must not be used)
            cuts = sorted(set(int(x) for x in rng.integers(1, n, 5))) + [n]
            lo = 0
            for k, hi in enumerate(cuts):                                       # text up to record hi, then rows lo..hi (or the other way round)
                a1, a2, a3 = np.ascontiguousarray(wrong[lo:hi]), np.ascontiguousarray(stop[lo:hi]), (~late[lo:hi]).astype(np.uint8)
                steps = [lambda: L.itsx_twriter_text(w, base, int(ends[hi - 1]), 1 if hi == n else 0),
                         lambda: L.itsx_twriter_coords(w, lo, hi - lo, a1.ctypes.data, a2.ctypes.data, a3.ctypes.data)]
                for f in (steps if (k + trial) % 2 else steps[::-1]):
                    assert f() == 0, L.itsx_trim_last_error()
                lo = hi
            idx = np.flatnonzero(late).astype(np.int64)
            if trial == 2:                                                      # close before the late rows are named: refused, nothing hangs
                nw, tot = C.c_int64(), C.c_int64()
                assert L.itsx_twriter_close(w, C.byref(nw), C.byref(tot)) == -1
                assert b"never decided" in L.itsx_trim_last_error()
                continue
            for part in np.array_split(idx, 3):
                s1, s2 = np.ascontiguousarray(start[part]), np.ascontiguousarray(stop[part])
                assert L.itsx_twriter_update(w, part.ctypes.data, len(part), s1.ctypes.data, s2.ctypes.data) == 0
            nw, tot = C.c_int64(), C.c_int64()
            assert L.itsx_twriter_close(w, C.byref(nw), C.byref(tot)) == 0, L.itsx_trim_last_error()
            assert (nw.value, tot.value) == tuple(want)
            assert out.read_bytes() == ref.read_bytes(), (kind, trial)
    # coordinates out of order are refused
    w = C.c_void_p()
    assert L.itsx_twriter_open(os.fsencode(str(tmp_path / "x")), 0, 0, C.byref(w)) == 0
    assert L.itsx_twriter_coords(w, 5, 1, start.ctypes.data, stop.ctypes.data, None) == -1
    L.itsx_twriter_text(w, None, 0, 1)
    assert L.itsx_twriter_close(w, None, None) == 0


def test_writer_object_writes_a_held_back_file_in_one_burst(tmp_path, monkeypatch):
    """the FIRST record is decided last (a streaming run's exact thresholds): every unit behind it is finished and waits, then 40 MB go
    out at once -- at explicit offsets, on several threads -- and the file is still the one-go writer's, byte for byte"""
    import ctypes as C
    from itsxpress_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(9)
    n, ln = 130000, 150
    acgt, ql = np.frombuffer(b"ACGT", np.uint8), np.frombuffer(b"#5FI", np.uint8)
    sq, qq = acgt[rng.integers(0, 4, (n, ln))], ql[rng.integers(0, 4, (n, ln))]
    text = b"".join(b"@r%d x\n" % i + sq[i].tobytes() + b"\n+\n" + qq[i].tobytes() + b"\n" for i in range(n))
    fq = tmp_path / "in.fq"
    fq.write_bytes(text)
    start, stop = np.zeros(n, np.int32), np.full(n, ln, np.int32)
    monkeypatch.setenv("ITSX_WRITE_UNIT_KB", "1024")
    monkeypatch.setenv("ITSX_IO_THREADS", "6")
    ref = tmp_path / "ref.fq"
    want = write_trimmed_fastq(str(fq), str(ref), start, stop)
    assert ref.stat().st_size > (36 << 20)
    out = tmp_path / "burst.fq"
    w = C.c_void_p()
    assert L.itsx_twriter_open(os.fsencode(str(out)), 0, 0, C.byref(w)) == 0
    buf = C.create_string_buffer(text, len(text))
    assert L.itsx_twriter_text(w, C.addressof(buf), len(text), 1) == 0
    dec = np.ones(n, np.uint8); dec[0] = 0
    wrong = start.copy(); wrong[0] = 99
    assert L.itsx_twriter_coords(w, 0, n, wrong.ctypes.data, stop.ctypes.data, dec.ctypes.data) == 0
    import time
    time.sleep(1.0)                                               # (the pool finishes every unit but the first)
    assert out.stat().st_size == 0
    rec = np.zeros(1, np.int64)
    assert L.itsx_twriter_update(w, rec.ctypes.data, 1, start.ctypes.data, stop.ctypes.data) == 0
    nw, tot = C.c_int64(), C.c_int64()
    assert L.itsx_twriter_close(w, C.byref(nw), C.byref(tot)) == 0, L.itsx_trim_last_error()
    assert (nw.value, tot.value) == tuple(want)
    assert out.read_bytes() == ref.read_bytes()


def test_single_end_writer_into_a_pipe(tmp_path, monkeypatch):
    """the output may be a pipe (a process substitution, /dev/stdout): no offsets there, the pieces go out one after the other"""
    import threading
    rng = np.random.default_rng(10)
    n = 3000
    fq = tmp_path / "in.fq"
    with open(fq, "w") as f:
        for i in range(n):
            ln = int(rng.integers(40, 300))
            f.write("@p%d\n%s\n+\n%s\n" % (i, "".join(rng.choice(list("ACGT"), ln)), "I" * ln))
    start, stop = rng.integers(0, 20, n).astype(np.int32), rng.integers(20, 300, n).astype(np.int32)
    monkeypatch.setenv("ITSX_WRITE_UNIT_KB", "64")
    monkeypatch.setenv("ITSX_IO_THREADS", "4")
    ref = tmp_path / "ref.fq.gz"
    want = write_trimmed_fastq(str(fq), str(ref), start, stop, gzipped=True)
    fifo = tmp_path / "out.fifo"
    os.mkfifo(fifo)
    got = []
    t = threading.Thread(target=lambda: got.append(open(fifo, "rb").read()))
    t.start()
    assert write_trimmed_fastq(str(fq), str(fifo), start, stop, gzipped=True) == want
    t.join(timeout=30)
    assert gzip.decompress(got[0]) == gzip.decompress(ref.read_bytes())


def test_parallel_paired_writer_writes_the_same_bytes(tmp_path, monkeypatch):
    """large R1 / R2 are cut at the same record numbers and sliced by a pool of threads (R2's records have other sizes than R1's: a
    range of R2 starts inside another of its byte ranges); labels go through a hash index; same bytes as the serial walk, also when
    R2 is the longer file, when labels repeat and when most pairs are dropped"""
    rng = np.random.default_rng(21)
    n = 7000
    r1, r2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    names = []
    with open(r1, "w") as f1, open(r2, "w") as f2:
        for i in range(n + 50):                                   # R2 holds 50 records more: zip() stops at R1's end
            l1, l2 = int(rng.integers(30, 260)), int(rng.integers(200, 400))
            nm = "M0:%d:%d" % (i // 100, i % 100)
            q1 = "".join(rng.choice(list("@+IF#5"), l1)); q2 = "".join(rng.choice(list("@+IF#5"), l2))
            if i < n:
                f1.write("@%s 1:N:0\n%s\n+\n%s\n" % (nm, "".join(rng.choice(list("ACGT"), l1)), q1))
                names.append(nm)
            f2.write("@%s 2:N:0\n%s\n+\n%s\n" % (nm, "".join(rng.choice(list("ACGT"), l2)), q2))
    keep = np.flatnonzero(rng.random(n) < 0.8)                    # the merged reads: a subsequence of the pairs
    mnames = [names[i] for i in keep] + [names[int(keep[0])]]     # (one label twice: the first entry counts)
    m = len(mnames)
    start = rng.integers(-1, 60, m).astype(np.int32)
    stop = rng.integers(-1, 300, m).astype(np.int32)
    tlen = rng.integers(250, 480, m).astype(np.int32)
    out = {}
    for mode, env in (("serial", {"ITSX_IO_THREADS": "1"}), ("parallel", {"ITSX_IO_THREADS": "5", "ITSX_WRITE_MIN_MB": "0", "ITSX_WRITE_UNIT_KB": "48"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for kind, kw in (("plain", {}), ("ccs", {"trim_ccs": True})):
            o1, o2 = tmp_path / ("%s_%s_1.fq" % (mode, kind)), tmp_path / ("%s_%s_2.fq" % (mode, kind))
            nw = write_trimmed_paired(str(r1), str(r2), str(o1), str(o2), mnames, start, stop, tlen, **kw)
            out[(mode, kind)] = (nw, o1.read_bytes(), o2.read_bytes())
        o1, o2 = tmp_path / (mode + "_1.fq.gz"), tmp_path / (mode + "_2.fq.gz")
        nw = write_trimmed_paired(str(r1), str(r2), str(o1), str(o2), mnames, start, stop, tlen, gzipped=True)
        out[(mode, "gz")] = (nw, gzip.decompress(o1.read_bytes()), gzip.decompress(o2.read_bytes()))
        for k in env:
            monkeypatch.delenv(k)
    for kind in ("plain", "ccs", "gz"):
        assert out[("serial", kind)] == out[("parallel", kind)], kind
    assert out[("serial", "gz")][1:] == out[("serial", "plain")][1:] and out[("serial", "plain")][0] > 1000
    # a malformed record in R2 is still reported
    with open(r2, "a") as f2:
        pass
    bad = tmp_path / "bad2.fq"
    bad.write_text(open(r2).read().replace("\n+\n", "\n-\n", 1))
    monkeypatch.setenv("ITSX_WRITE_MIN_MB", "0")
    monkeypatch.setenv("ITSX_WRITE_UNIT_KB", "48")
    with pytest.raises(EngineError) as e:
        write_trimmed_paired(str(r1), str(bad), str(tmp_path / "x1"), str(tmp_path / "x2"), mnames, start, stop, tlen)
    assert "malformed FASTQ record" in str(e.value)


def test_label_index_built_by_a_pool_keeps_the_first_of_equal_labels(tmp_path, monkeypatch):
    """70 k merged labels (the index is filled by several threads from 65 536 on), every tenth label a second time with OTHER
    coordinates: the first entry counts, whatever thread inserted which first; and a sample nothing of which survives is still
    one valid (empty) gzip member per file"""
    rng = np.random.default_rng(22)
    n = 70000
    r1, r2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    with open(r1, "w") as f1, open(r2, "w") as f2:
        for i in range(n):
            f1.write("@r%d 1\n%s\n+\n%s\n" % (i, "ACGTACGTACGTACGTACGTACGTACGTAC", "I" * 30))
            f2.write("@r%d 2\n%s\n+\n%s\n" % (i, "TTGCATTGCATTGCATTGCATTGCATTGCA", "F" * 30))
    first = ["r%d" % i for i in range(n)]
    dup = ["r%d" % i for i in range(0, n, 10)]
    perm = rng.permutation(len(dup))
    mnames = first + [dup[i] for i in perm]
    m = len(mnames)
    start = np.concatenate([rng.integers(0, 10, n), np.full(len(dup), 29)]).astype(np.int32)
    stop = np.concatenate([rng.integers(10, 30, n), np.full(len(dup), 30)]).astype(np.int32)
    tlen = np.full(m, 30, np.int32)
    out = {}
    for mode, env in (("serial", {"ITSX_IO_THREADS": "1"}), ("pool", {"ITSX_IO_THREADS": "6", "ITSX_WRITE_MIN_MB": "0", "ITSX_WRITE_UNIT_KB": "256"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        o1, o2 = tmp_path / (mode + "_1.fq.gz"), tmp_path / (mode + "_2.fq.gz")
        nw = write_trimmed_paired(str(r1), str(r2), str(o1), str(o2), mnames, start, stop, tlen, gzipped=True)
        out[mode] = (nw, gzip.decompress(o1.read_bytes()), gzip.decompress(o2.read_bytes()))
        e1, e2 = tmp_path / (mode + "_e1.fq.gz"), tmp_path / (mode + "_e2.fq.gz")
        assert write_trimmed_paired(str(r1), str(r2), str(e1), str(e2), mnames, np.full(m, -1, np.int32), stop, tlen, gzipped=True) == 0
        assert gzip.decompress(e1.read_bytes()) == b"" and gzip.decompress(e2.read_bytes()) == b""
        for k in env:
            monkeypatch.delenv(k)
    assert out["serial"] == out["pool"] and out["pool"][0] == n
    recs = out["pool"][1].split(b"\n")
    for i in (0, 10, 69990):                                  # a repeated label: its FIRST coordinates
        assert len(recs[4 * i + 1]) == int(stop[i] - start[i]), i


def test_writer_object_in_slice_mode_follows_python_slicing(tmp_path):
    """itsx_twriter_set_mode(w, 1) (round 6: the two mates of a streamed paired sample, itsxpress/SeqSample.py:587-670): (start, stop) are Python
    slice bounds as they come -- negative starts count from the end, stop == INT32_MAX is an open end, stop == INT32_MIN skips the record;
    what the paired writer's own arithmetic gives, R1[start:stop] (or [start:]) and R2[tlen - stop : tlen - start] (or [.. :]), must come
    out of two such writers exactly as out of itsx_write_trimmed_paired"""
    import ctypes as C
    from itsxpress_amd import _lib
    rng = np.random.default_rng(5)
    n = 3000
    acgt = np.frombuffer(b"ACGT", np.uint8)
    recs1, recs2, names, start, stop, tlen = [], [], [], [], [], []
    for i in range(n):
        l1, l2 = int(rng.integers(60, 250)), int(rng.integers(60, 250))
        s1, s2 = acgt[rng.integers(0, 4, l1)].tobytes(), acgt[rng.integers(0, 4, l2)].tobytes()
        q1, q2 = bytes(rng.integers(35, 74, l1).astype(np.uint8)), bytes(rng.integers(35, 74, l2).astype(np.uint8))
        recs1.append(b"@p%d 1:N\n" % i + s1 + b"\n+\n" + q1 + b"\n")
        recs2.append(b"@p%d 2:N\n" % i + s2 + b"\n+\n" + q2 + b"\n")
        names.append("p%d" % i)
        t = int(rng.integers(120, 480))
        a = int(rng.integers(0, 150))
        b = int(rng.integers(a - 5, t + 40))            # also stop <= start (skipped) and stop > tlen (open end, negative R2 start)
        if i % 17 == 0:
            a, b = -1, -1                                # not trimmed
        start.append(a); stop.append(b); tlen.append(t)
    t1, t2 = b"".join(recs1), b"".join(recs2)
    f1, f2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    f1.write_bytes(t1); f2.write_bytes(t2)
    start, stop, tlen = (np.asarray(x, np.int32) for x in (start, stop, tlen))
    o1, o2 = tmp_path / "o1.fq", tmp_path / "o2.fq"
    nw = write_trimmed_paired(str(f1), str(f2), str(o1), str(o2), names, start, stop, tlen)
    # the same through two writer objects in slice mode
    SKIP, OPEN = -(1 << 31), (1 << 31) - 1
    s64, e64, t64 = start.astype(np.int64), stop.astype(np.int64), tlen.astype(np.int64)
    keep = (s64 >= 0) & (e64 >= 0) & (s64 < e64)
    a1 = np.where(keep, s64, 0).astype(np.int32); b1 = np.where(keep, np.where(e64 > t64, OPEN, e64), SKIP).astype(np.int32)
    a2 = np.where(keep, t64 - e64, 0).astype(np.int32); b2 = np.where(keep, np.where(t64 - s64 > t64, OPEN, t64 - s64), SKIP).astype(np.int32)
    L = _lib.lib()
    got = []
    for text, a, b, name in ((t1, a1, b1, "w1.fq"), (t2, a2, b2, "w2.fq")):
        w = C.c_void_p()
        out = tmp_path / name
        assert L.itsx_twriter_open(os.fsencode(str(out)), 0, 0, C.byref(w)) == 0
        assert L.itsx_twriter_set_mode(w, 1) == 0
        buf = C.create_string_buffer(text, len(text))
        dec = np.ones(n, np.uint8)
        assert L.itsx_twriter_text(w, buf, len(text), 1) == 0
        assert L.itsx_twriter_coords(w, 0, n, a.ctypes.data, b.ctypes.data, dec.ctypes.data) == 0
        cnt, tot = C.c_int64(0), C.c_int64(0)
        assert L.itsx_twriter_close(w, C.byref(cnt), C.byref(tot)) == 0, L.itsx_trim_last_error()
        assert cnt.value == nw == int(keep.sum())
        got.append(out.read_bytes())
    assert got[0] == o1.read_bytes() and got[1] == o2.read_bytes() and len(got[0]) > 10000
    w = C.c_void_p()
    assert L.itsx_twriter_open(os.fsencode(str(tmp_path / "x.fq")), 0, 0, C.byref(w)) == 0
    assert L.itsx_twriter_set_mode(w, 2) != 0
    L.itsx_twriter_text(w, None, 0, 1)
    L.itsx_twriter_close(w, None, None)
