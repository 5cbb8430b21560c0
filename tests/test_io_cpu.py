"""Host-side file plumbing: the reader behind every *_file entry point and the block-parallel
gzip / zstd writers.  CPU-only; the content written must be byte-identical to a single-stream writer's."""
import gzip
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

from itsxpress_amd import _lib
from itsxpress_amd._lib import EngineError
from itsxpress_amd.engine import read_fastx
from itsxpress_amd.trim import read_text, write_trimmed_fastq, write_trimmed_paired

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fastq(n, seed, lo=40, hi=400):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        s = "".join(rng.choice(list("ACGTN"), L, p=[.245, .245, .245, .245, .02]))
        q = "".join(chr(33 + int(x)) for x in rng.integers(2, 41, L))
        recs.append("@r%d extra words\n%s\n+\n%s\n" % (i, s, q))
    return "".join(recs)


def _expected(text, start, stop):
    lines = text.split("\n")
    out = []
    for i in range(len(start)):
        t, s, q = lines[4 * i], lines[4 * i + 1], lines[4 * i + 3]
        a, b = int(start[i]), int(stop[i])
        if a < 0 or b < 0 or not a < b:
            continue
        out.append("%s\n%s\n+\n%s\n" % (t, s[a:b], q[a:b]))
    return "".join(out)


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    d = tmp_path_factory.mktemp("io")
    text = _fastq(3000, 5)
    rng = np.random.default_rng(6)
    start = rng.integers(-1, 60, 3000).astype(np.int32)
    stop = rng.integers(20, 420, 3000).astype(np.int32)
    plain = d / "in.fastq"
    plain.write_text(text)
    return d, text, start, stop, str(plain)


def test_codecs_are_reported():
    flags = _lib.lib().itsx_io_codecs()
    assert flags & 2, "libzstd.so.1 is part of this image"


def test_reader_takes_plain_gzip_multimember_and_zstd(sample):
    d, text, start, stop, plain = sample
    raw = text.encode()
    one = d / "one.gz"
    one.write_bytes(gzip.compress(raw, 1))
    multi = d / "multi.gz"                     # concatenated members, as bgzip / pigz / cat a.gz b.gz produce
    cut = [0, 1000, 1001, len(raw) // 2, len(raw)]
    multi.write_bytes(b"".join(gzip.compress(raw[cut[i]:cut[i + 1]], 6) for i in range(4)) + gzip.compress(b""))
    assert read_text(plain) == raw and read_text(str(one)) == raw and read_text(str(multi)) == raw
    zst = d / "out.zst"
    n, _ = write_trimmed_fastq(plain, str(zst), np.zeros(3000, np.int32), np.full(3000, 10000, np.int32), zstd_file=True)
    assert n == 3000 and zst.read_bytes()[:4] == b"\x28\xb5\x2f\xfd"
    assert read_text(str(zst)) == raw          # stop beyond the read clamps: the whole record
    names, seqs = read_fastx(str(zst))
    assert names[:2] == ["r0", "r1"] and len(seqs) == 3000 and seqs[7] == text.split("\n")[4 * 7 + 1]
    empty = d / "empty.fq"
    empty.write_bytes(b"")
    assert read_text(str(empty)) == b"" and read_fastx(str(empty)) == ([], [])


@pytest.mark.parametrize("env", [{"ITSX_IO_BLOCK_KB": "16", "ITSX_WRITE_UNIT_KB": "24", "ITSX_IO_THREADS": "5"},
                                 {"ITSX_IO_BLOCK_KB": "16", "ITSX_WRITE_UNIT_KB": "24", "ITSX_IO_THREADS": "3", "ITSX_IO_LIBDEFLATE": "0"},
                                 {"ITSX_IO_THREADS": "1"}])
def test_block_parallel_output_is_one_stream_to_any_reader(sample, env):
    """Many small blocks, several threads, both deflate implementations: Python's gzip / zlib must read back exactly
    what the plain writer wrote (run in a child process: the knobs are read when the library first needs them)."""
    d, text, start, stop, plain = sample
    code = r"""
import sys, gzip, zlib, numpy as np
sys.path.insert(0, %r)
from itsxpress_amd.trim import write_trimmed_fastq, read_text
from itsxpress_amd import _lib
d, plain = %r, %r
start = np.load(d + '/start.npy'); stop = np.load(d + '/stop.npy')
a = write_trimmed_fastq(plain, d + '/o.fq', start, stop)
b = write_trimmed_fastq(plain, d + '/o.fq.gz', start, stop, gzipped=True)
c = write_trimmed_fastq(plain, d + '/o.fq.zst', start, stop, zstd_file=True)
assert a == b == c, (a, b, c)
want = open(d + '/o.fq', 'rb').read()
gz = open(d + '/o.fq.gz', 'rb').read()
assert gzip.decompress(gz) == want
members = 0; rest = gz
while rest:
    o = zlib.decompressobj(31); o.decompress(rest); rest = o.unused_data; members += 1
assert read_text(d + '/o.fq.gz') == want and read_text(d + '/o.fq.zst') == want
print(members, _lib.lib().itsx_io_codecs(), a[0])
""" % (ROOT, str(d), plain)
    np.save(str(d / "start.npy"), start)
    np.save(str(d / "stop.npy"), stop)
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    members, flags, nrec = map(int, r.stdout.split())
    assert open(str(d / "o.fq")).read() == _expected(text, start, stop) and nrec > 2000
    if "ITSX_IO_BLOCK_KB" in env:
        assert members > 20                    # ~1 MB of output, one member per 24 KB of input
    else:
        assert members == 1
    if env.get("ITSX_IO_LIBDEFLATE") == "0":
        assert flags & 1 == 0


def test_empty_outputs_are_valid_files(sample, tmp_path):
    d, text, start, stop, plain = sample
    none = np.full(3000, -1, np.int32)
    g, z, p = str(tmp_path / "e.gz"), str(tmp_path / "e.zst"), str(tmp_path / "e.fq")
    assert write_trimmed_fastq(plain, g, none, none, gzipped=True) == (0, 0)
    assert write_trimmed_fastq(plain, z, none, none, zstd_file=True) == (0, 0)
    assert write_trimmed_fastq(plain, p, none, none) == (0, 0)
    assert gzip.open(g).read() == b"" and read_text(z) == b"" and os.path.getsize(p) == 0 and os.path.getsize(z) > 0


def test_rewritten_input_is_not_served_from_the_text_cache(tmp_path):
    """The writers share the loader's decompressed text through a cache keyed on (path, size, mtime)."""
    f = str(tmp_path / "x.fastq.gz")
    a, b = _fastq(50, 1), _fastq(50, 2)
    st, sp = np.zeros(50, np.int32), np.full(50, 1000, np.int32)
    for text in (a, b, a):
        with gzip.open(f, "wt") as g:
            g.write(text)
        out = str(tmp_path / "o.fq")
        assert write_trimmed_fastq(f, out, st, sp)[0] == 50
        assert open(out).read() == text.replace(" extra words\n", " extra words\n")


def test_corrupt_and_missing_inputs_fail_loudly(sample, tmp_path):
    d, text, start, stop, plain = sample
    raw = gzip.compress(text.encode())
    bad = tmp_path / "trunc.gz"
    bad.write_bytes(raw[: len(raw) // 2])
    with pytest.raises(EngineError):
        read_text(str(bad))
    with pytest.raises(EngineError):
        write_trimmed_fastq(str(bad), str(tmp_path / "o.fq"), start, stop)
    flipped = bytearray(raw)
    flipped[len(raw) // 2] ^= 0xFF
    (tmp_path / "flip.gz").write_bytes(bytes(flipped))
    with pytest.raises(EngineError):
        read_text(str(tmp_path / "flip.gz"))
    with pytest.raises(FileNotFoundError):
        read_text(str(tmp_path / "nope.fq"))
    with pytest.raises(EngineError):           # unwritable destination
        write_trimmed_fastq(plain, str(tmp_path / "no_such_dir" / "o.fq.gz"), start, stop, gzipped=True)


def test_paired_writer_compressed_outputs_match_plain(tmp_path):
    rng = np.random.default_rng(3)
    t1, t2 = _fastq(400, 11, 100, 250), _fastq(400, 12, 100, 250)
    f1, f2 = tmp_path / "a.fastq.gz", tmp_path / "b.fastq.gz"
    f1.write_bytes(gzip.compress(t1.encode()))
    f2.write_bytes(gzip.compress(t2.encode()))
    names = ["r%d" % i for i in range(400)]
    start = rng.integers(0, 40, 400).astype(np.int32)
    stop = rng.integers(60, 300, 400).astype(np.int32)
    tlen = rng.integers(200, 320, 400).astype(np.int32)
    outs = {}
    for kind, kw in (("plain", {}), ("gz", {"gzipped": True}), ("zst", {"zstd_file": True})):
        o1, o2 = str(tmp_path / ("o1." + kind)), str(tmp_path / ("o2." + kind))
        assert write_trimmed_paired(str(f1), str(f2), o1, o2, names, start, stop, tlen, **kw) == 400
        outs[kind] = (read_text(o1), read_text(o2))
    assert outs["plain"] == outs["gz"] == outs["zst"] and len(outs["plain"][0]) > 10000
    assert open(str(tmp_path / "o1.gz"), "rb").read(2) == b"\x1f\x8b"


def test_temp_files_the_next_stage_reads_back_come_from_the_cache_and_stay_honest(tmp_path):
    """oriented.fq / seq.fq are written with their text left in the reader's cache; the key is (path, size, mtime), so a
    file changed behind the library's back is read again."""
    from itsxpress_amd.trim import write_oriented_fastq
    text = _fastq(200, 21, 60, 61)
    src = tmp_path / "in.fq"
    src.write_text(text)
    strand = np.ones(200, np.int8)
    strand[::3] = -1
    strand[5] = 0
    ori = str(tmp_path / "oriented.fq")
    assert write_oriented_fastq(str(src), ori, strand) == 199
    on_disk = open(ori).read()
    out = str(tmp_path / "o.fq")
    st, sp = np.zeros(199, np.int32), np.full(199, 1000, np.int32)
    assert write_trimmed_fastq(ori, out, st, sp)[0] == 199 and open(out).read() == on_disk
    swapped = on_disk.replace("A", "x").replace("C", "A").replace("x", "C")       # same length, other content
    assert swapped != on_disk
    with open(ori, "w") as f:
        f.write(swapped)
    assert write_trimmed_fastq(ori, out, st, sp)[0] == 199 and open(out).read() == swapped


# ---------------------------------------------------------------------------------------------------------------------
# block-parallel inflate of single-member gzip files (csrc/pinflate.cpp)
def _amplicon_fastq(n, seed):
    """FASTQ text with the redundancy of an amplicon run (so that back-references reach far back, across chunk borders)"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    tmpl = acgt[rng.integers(0, 4, (max(1, n // 40), 300))]
    reads = tmpl[rng.integers(0, len(tmpl), n)].copy()
    err = rng.random(reads.shape) < 0.003
    reads[err] = acgt[rng.integers(0, 4, int(err.sum()))]
    q = (np.clip(38 - (np.arange(300) // 25)[None, :] - rng.integers(0, 6, reads.shape), 2, 40) + 33).astype(np.uint8)
    return b"".join(b"@read%d 1:N:0:1\n" % i + reads[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n" for i in range(n))


@pytest.mark.parametrize("level", [1, 6, 9])
def test_parallel_inflate_equals_the_serial_inflaters(tmp_path, monkeypatch, level):
    text = _amplicon_fastq(12000, 40 + level)                     # 7.5 MB of text
    p = tmp_path / ("in%d.fastq.gz" % level)
    p.write_bytes(gzip.compress(text, level))
    L = _lib.lib()
    monkeypatch.setenv("ITSX_TEXT_CACHE_GB", "0")
    monkeypatch.setenv("ITSX_IO_THREADS", "5")
    monkeypatch.setenv("ITSX_PINFLATE_CHUNK_KB", "100")           # ~30 chunks, several rounds of 5
    before = L.itsx_io_parallel_inflates()
    assert read_text(str(p)) == text
    assert L.itsx_io_parallel_inflates() == before + 1, "the block-parallel inflater did not deliver this file"
    monkeypatch.setenv("ITSX_PARALLEL_INFLATE", "0")
    assert read_text(str(p)) == text
    assert L.itsx_io_parallel_inflates() == before + 1


def test_parallel_inflate_chunk_sizes_and_thread_counts(tmp_path, monkeypatch):
    text = _amplicon_fastq(9000, 50)
    p = tmp_path / "in.fastq.gz"
    p.write_bytes(gzip.compress(text, 6))
    monkeypatch.setenv("ITSX_TEXT_CACHE_GB", "0")
    L = _lib.lib()
    for threads, kb in ((2, 500), (3, 64), (8, 16), (7, 37), (4, 1000)):
        monkeypatch.setenv("ITSX_IO_THREADS", str(threads))
        monkeypatch.setenv("ITSX_PINFLATE_CHUNK_KB", str(kb))
        before = L.itsx_io_parallel_inflates()
        assert read_text(str(p)) == text, (threads, kb)
        # 1000-KB chunks: the file is smaller than four chunks and goes to the serial inflater
        assert L.itsx_io_parallel_inflates() == before + (0 if kb == 1000 else 1), (threads, kb)


def test_parallel_inflate_multi_member_binary_stored_and_damaged(tmp_path, monkeypatch):
    """Files of several members (two big ones; hundreds of small ones as bgzip / this package's own writers make them; an
    empty member in between), a binary head, and stored blocks are read correctly by the pool; bytes behind the last
    member send the file to the serial inflater; damaged data is rejected loudly.  The parallel result is only ever
    accepted when every member's CRC-32 and length match its trailer."""
    monkeypatch.setenv("ITSX_TEXT_CACHE_GB", "0")
    monkeypatch.setenv("ITSX_IO_THREADS", "4")
    monkeypatch.setenv("ITSX_PINFLATE_CHUNK_KB", "64")
    L = _lib.lib()
    text = _amplicon_fastq(6000, 60)
    half = len(text) // 2
    small = b"".join(gzip.compress(text[i:i + 50000], 6) for i in range(0, len(text), 50000))
    two_sym = np.random.default_rng(3).integers(65, 67, 24 << 20, dtype=np.uint8).tobytes()
    cases = {
        "two_members": (gzip.compress(text[:half], 6) + gzip.compress(b"", 6) + gzip.compress(text[half:], 1), text, 1),
        "small_members": (small, text, 1),
        "binary_head": (gzip.compress(np.random.default_rng(1).integers(0, 256, 3 << 20, dtype=np.uint8).tobytes() + text, 6), None, 1),
        "stored": (gzip.compress(text, 0), text, 1),
        "zeros_behind": (gzip.compress(text, 6) + b"\0" * 64, text, 0),
        # eight times its compressed size, in many members: the output outgrows the first guess (4 x) while the pool fills it
        "outgrows_the_guess": (b"".join(gzip.compress(two_sym[i:i + (1 << 20)], 6) for i in range(0, len(two_sym), 1 << 20)), two_sym, 1),
    }
    for name, (blob, want, parallel) in cases.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(blob)
        before = L.itsx_io_parallel_inflates()
        got = read_text(str(p))
        assert got == (want if want is not None else gzip.decompress(blob)), name
        assert L.itsx_io_parallel_inflates() == before + parallel, name
    # a flipped bit in the middle of the stream, a wrong CRC in a trailer (last and inner member): loud errors, never silent text
    good = gzip.compress(text, 6)
    two = gzip.compress(text[:half], 6)
    for name, blob in (("flipped", good[:len(good) // 2] + bytes([good[len(good) // 2] ^ 0x10]) + good[len(good) // 2 + 1:]),
                       ("trailer", good[:-8] + bytes([good[-8] ^ 1]) + good[-7:]),
                       ("inner_trailer", two[:-8] + bytes([two[-8] ^ 1]) + two[-7:] + gzip.compress(text[half:], 6))):
        p = tmp_path / (name + ".gz")
        p.write_bytes(blob)
        with pytest.raises(EngineError):
            read_text(str(p))


def _stream_slices(path, min_bytes, keep=0):
    import ctypes as C
    L = _lib.lib()
    h = C.c_void_p()
    rc = L.itsx_stream_open(os.fsencode(str(path)), C.byref(h))
    if rc != 0:
        raise EngineError(rc, L.itsx_stream_last_error().decode())
    out = []
    try:
        while True:
            ptr, nb, last = C.c_void_p(), C.c_int64(0), C.c_int32(0)
            rc = L.itsx_stream_next(h, min_bytes, C.byref(ptr), C.byref(nb), C.byref(last))
            if rc != 0:
                raise EngineError(rc, L.itsx_stream_last_error().decode())
            out.append(C.string_at(ptr.value, nb.value) if nb.value else b"")
            if last.value:
                break
    except BaseException:
        L.itsx_stream_close(h, 0)
        raise
    rc = L.itsx_stream_close(h, keep)
    if rc != 0:
        raise EngineError(rc, L.itsx_stream_last_error().decode())
    return out


def test_text_stream_hands_out_whole_records_while_inflating(tmp_path, monkeypatch):
    """itsx_stream_*: the slices of a large gzip FASTQ are cut at record starts (qualities that start with '@' included), arrive
    before the file is done and add up to the serial inflater's text; plain, zstd-less and small files arrive in one slice; the
    text lands in the cache for the writers; a damaged file is an error of the last call at the latest."""
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    recs = []
    for i in range(40000):
        ln = int(rng.integers(40, 420))
        q = (rng.integers(0, 41, ln) + 33).astype(np.uint8)
        if i % 3 == 0:
            q[0] = ord("@")
        recs.append(b"@r%d x\n" % i + acgt[rng.integers(0, 4, ln)].tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    text = b"".join(recs)
    p = tmp_path / "in.fastq.gz"
    p.write_bytes(gzip.compress(text, 6))
    L = _lib.lib()
    monkeypatch.setenv("ITSX_IO_THREADS", "4")
    monkeypatch.setenv("ITSX_PINFLATE_CHUNK_KB", "64")
    monkeypatch.setenv("ITSX_TEXT_CACHE_GB", "1")
    L.itsx_io_cache_clear()
    before = L.itsx_io_parallel_inflates()
    sl = _stream_slices(p, 1 << 20, keep=1)
    assert L.itsx_io_parallel_inflates() == before + 1
    assert len(sl) >= 5 and b"".join(sl) == text
    assert all(s[:1] == b"@" and s.count(b"\n") % 4 == 0 and s.endswith(b"\n") for s in sl)
    # every slice parses on its own to the records it holds
    from itsxpress_amd.trim import read_names as fastq_ids
    n = 0
    for k, s in enumerate(sl[:3]):
        q = tmp_path / ("slice%d.fq" % k)
        q.write_bytes(s)
        ids = fastq_ids(str(q))
        assert ids[0] == "r%d" % n and len(ids) == s.count(b"\n") // 4
        n += len(ids)
    # kept: the text is in the loaders' / writers' cache -- a second stream of the path is cut from it, nothing is inflated again
    again = _stream_slices(p, 1 << 20)
    assert again == sl or (b"".join(again) == text and all(len(x) <= (3 << 19) + (1 << 18) for x in again[:-1]))
    assert L.itsx_io_parallel_inflates() == before + 1
    assert all((1 << 20) <= len(x) <= (3 << 19) + (1 << 18) for x in sl[:-1]), [len(x) for x in sl]
    L.itsx_io_cache_clear()
    # text that is final at once (a plain file; the serial inflater, ITSX_PARALLEL_INFLATE=0) is cut the same way; a small file
    # and FASTA (never cut) arrive in one piece
    plain = tmp_path / "in.fastq"
    plain.write_bytes(text)
    pl = _stream_slices(plain, 1 << 20)
    assert len(pl) >= 5 and b"".join(pl) == text and all(x[:1] == b"@" and x.count(b"\n") % 4 == 0 for x in pl)
    small = tmp_path / "small.fastq.gz"
    small.write_bytes(gzip.compress(text[:100000], 6))
    assert _stream_slices(small, 1 << 20) == [text[:100000]]
    tiny = _stream_slices(small, 1 << 10)                                    # (slices of 1 - 1.5 KB + the cut window: tests use them)
    assert len(tiny) > 10 and b"".join(tiny) == text[:100000] and all(x[:1] == b"@" for x in tiny)
    fa = tmp_path / "in.fa.gz"
    fasta = b"".join(b">s%d\n" % i + acgt[rng.integers(0, 4, 300)].tobytes() + b"\n" for i in range(30000))
    fa.write_bytes(gzip.compress(fasta, 6))
    assert _stream_slices(fa, 1 << 20) == [fasta]
    monkeypatch.setenv("ITSX_PARALLEL_INFLATE", "0")
    assert b"".join(_stream_slices(p, 1 << 20)) == text
    monkeypatch.delenv("ITSX_PARALLEL_INFLATE")
    empty = tmp_path / "empty.fastq"
    empty.write_bytes(b"")
    assert _stream_slices(empty, 1 << 20) == [b""]
    # damage: a flipped byte late in the stream, a wrong CRC
    good = p.read_bytes()
    for name, blob in (("flipped", good[:len(good) * 3 // 4] + bytes([good[len(good) * 3 // 4] ^ 0x10]) + good[len(good) * 3 // 4 + 1:]),
                       ("trailer", good[:-8] + bytes([good[-8] ^ 1]) + good[-7:])):
        q = tmp_path / (name + ".gz")
        q.write_bytes(blob)
        with pytest.raises(EngineError):
            _stream_slices(q, 1 << 20)
    with pytest.raises(EngineError):
        _stream_slices(tmp_path / "missing.gz", 1)


def test_keyset_first_holder_wins():
    """itsx_keyset_assign: a key's first holder (chunk, local unique, first occurrence, orientation) answers every later chunk"""
    import ctypes as C
    L = _lib.lib()
    ks = L.itsx_keyset_create()
    try:
        rng = np.random.default_rng(2)
        # (enough keys for the 16 sub-tables to grow several times, inside one call too: a slot noted before its table moved is found again)
        keys = rng.integers(-2**62, 2**62, (600000, 2), dtype=np.int64)
        seen = {}
        base = 0
        for chunk in range(4):
            pick = rng.integers(0, len(keys), 30000 if chunk == 0 else 350000)
            pick = pick[np.sort(np.unique(pick, return_index=True)[1])]          # a chunk's uniques are distinct
            tup = np.zeros((len(pick), 4), np.int64)
            tup[:, :2] = keys[pick]
            tup[:, 2] = base + np.arange(len(pick)) * 3
            tup[:, 3] = rng.integers(0, 2, len(pick))
            out = np.zeros_like(tup)
            gid = np.full(len(pick), -1, np.int64)
            assert L.itsx_keyset_assign(ks, tup.ctypes.data, len(pick), chunk, out.ctypes.data, gid.ctypes.data) == 0
            for u, k in enumerate(pick):
                want = seen.setdefault(int(k), (int(tup[u, 2]), int(tup[u, 3]), chunk, u, len(seen)))
                assert tuple(int(x) for x in out[u]) + (int(gid[u]),) == want          # gid: first-seen order
            base += 2000000
        assert L.itsx_keyset_size(ks) == len(seen)
        assert L.itsx_keyset_assign(ks, None, 0, 9, None, None) == 0
    finally:
        L.itsx_keyset_destroy(ks)


def test_text_stream_that_outgrows_its_reservation_says_so(tmp_path, monkeypatch):
    """slices must not move once they are out, so the stream reserves address space up front (64 x the file + 1 GB); a file that
    expands beyond it -- here the reservation is made too small on purpose -- is delivered by the serial inflater if nothing was
    handed out yet, and otherwise ends in an error that says so (the streaming driver then starts over with the serial inflater)"""
    text = _amplicon_fastq(20000, 77)                              # 12.5 MB, ~2.4 x its gzip
    p = tmp_path / "in.fastq.gz"
    p.write_bytes(gzip.compress(text, 6))
    L = _lib.lib()
    monkeypatch.setenv("ITSX_IO_THREADS", "4")
    monkeypatch.setenv("ITSX_PINFLATE_CHUNK_KB", "64")
    monkeypatch.setenv("ITSX_TEXT_CACHE_GB", "0")
    monkeypatch.setenv("ITSX_STREAM_RESERVE_X", "1")
    monkeypatch.setenv("ITSX_STREAM_RESERVE_MB", "4")
    with pytest.raises(EngineError) as e:                          # small slices: some are out when the room ends
        _stream_slices(p, 1 << 20)
    assert "gave up after slices" in str(e.value)
    # a first slice larger than the file: nothing is out when the room ends, the serial inflater takes over
    assert b"".join(_stream_slices(p, 1 << 30)) == text
    monkeypatch.setenv("ITSX_PARALLEL_INFLATE", "0")               # what the driver's second attempt does
    assert b"".join(_stream_slices(p, 1 << 20)) == text


def test_reader_takes_a_fifo(tmp_path):
    """a path without a size or positions (a FIFO, a process substitution) is read to its end like any other"""
    import threading
    text = _amplicon_fastq(3000, 5)
    blob = gzip.compress(text, 6)
    fifo = str(tmp_path / "reads.fq.gz")
    os.mkfifo(fifo)

    def feed():
        with open(fifo, "wb") as f:
            f.write(blob)

    t = threading.Thread(target=feed)
    t.start()
    try:
        assert read_text(fifo) == text
    finally:
        t.join()


def test_shared_text_stream_lets_another_mapping_read_the_slices(tmp_path, monkeypatch):
    """itsx_stream_open_shared (round 6; the multi-GPU driver's load: itsxpress_amd/multi.py): the text is inflated into a shared
    mapping of a file, a slice is readable through ANY mapping of that file at `address - base` as soon as it is handed out, the
    progress pair estimates the final size, the slices add up to the serial inflater's text; a plain input is not copied (its slices
    are offsets into the input itself)"""
    import ctypes as C
    import mmap
    monkeypatch.setenv("ITSX_PINFLATE_CHUNK_KB", "64")
    monkeypatch.setenv("ITSX_TEXT_CACHE_GB", "0")
    rng = np.random.default_rng(3)
    recs = []
    for i in range(60000):
        n = int(rng.integers(80, 160))
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n)), b"@" + b"I" * (n - 1)))
    text = b"".join(recs)
    gz = tmp_path / "in.fq.gz"
    with gzip.open(gz, "wb", compresslevel=6) as f:
        f.write(text)
    plain = tmp_path / "in.fq"
    plain.write_bytes(text)
    L = _lib.lib()
    for path, want_plain in ((gz, 0), (plain, 1)):
        backing = tmp_path / ("text_%d" % want_plain)
        h, pl = C.c_void_p(), C.c_int32(-1)
        assert L.itsx_stream_open_shared(os.fsencode(str(path)), os.fsencode(str(backing)), C.byref(h), C.byref(pl)) == 0, L.itsx_stream_last_error()
        assert pl.value == want_plain
        src = path if want_plain else backing
        base = L.itsx_stream_base(h)
        got, off = [], 0
        est = []
        while True:
            ptr, nb, last = C.c_void_p(), C.c_int64(0), C.c_int32(0)
            assert L.itsx_stream_next(h, 1 << 20, C.byref(ptr), C.byref(nb), C.byref(last)) == 0, L.itsx_stream_last_error()
            assert (ptr.value or base) - base == off
            a, c, r = C.c_int64(0), C.c_int64(0), C.c_int64(0)
            L.itsx_stream_progress(h, C.byref(a), C.byref(c), C.byref(r))
            if c.value:
                est.append(a.value * r.value / c.value)
            if nb.value:                                   # read the slice through a mapping of our own, like a worker process
                gran = mmap.ALLOCATIONGRANULARITY
                lo = (off // gran) * gran
                with open(src, "rb") as f:
                    mm = mmap.mmap(f.fileno(), nb.value + off - lo, access=mmap.ACCESS_READ, offset=lo)
                piece = bytes(mm[off - lo:])
                mm.close()
                assert piece[:1] == b"@"
                got.append(piece)
            off += nb.value
            if last.value:
                break
        assert L.itsx_stream_close(h, 0) == 0
        assert b"".join(got) == text
        assert len(got) > (1 if not want_plain else 0)
        assert all(0.7 * len(text) < e < 1.4 * len(text) for e in est), est[:4]
        if not want_plain:
            os.unlink(backing)


def test_mate_stream_hands_out_the_asked_number_of_records(tmp_path, monkeypatch):
    """itsx_stream_next_records / itsx_count_records / itsx_write_range (round 6: a streamed paired sample cuts R2 at R1's record counts; the
    multi-GPU driver writes a worker's piece while the rest of the file is inflated): slices hold exactly the records asked for, fewer
    only at the end of the file; a piece written from memory is the text it came from"""
    import ctypes as C
    monkeypatch.setenv("ITSX_PINFLATE_CHUNK_KB", "64")
    monkeypatch.setenv("ITSX_TEXT_CACHE_GB", "0")
    rng = np.random.default_rng(8)
    recs = []
    for i in range(50000):
        n = int(rng.integers(50, 200))
        recs.append(b"@m%d 2:N:0\n%s\n+\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n)), b"@" * n))
    text = b"".join(recs)
    gz = tmp_path / "r2.fq.gz"
    with gzip.open(gz, "wb", compresslevel=6) as f:
        f.write(text)
    L = _lib.lib()
    assert L.itsx_count_records(C.c_char_p(text), len(text)) == 50000
    assert L.itsx_count_records(C.c_char_p(text[:-1]), len(text) - 1) == 50000          # (a last line without its newline)
    h = C.c_void_p()
    assert L.itsx_stream_open(os.fsencode(str(gz)), C.byref(h)) == 0, L.itsx_stream_last_error()
    got, asked, off, n_left = [], [1, 7, 12000, 0, 20000, 30000], 0, 50000
    for want in asked:
        ptr, nb, cnt, last = C.c_void_p(), C.c_int64(0), C.c_int64(0), C.c_int32(0)
        assert L.itsx_stream_next_records(h, want, C.byref(ptr), C.byref(nb), C.byref(cnt), C.byref(last)) == 0, L.itsx_stream_last_error()
        piece = C.string_at(ptr.value, nb.value) if nb.value else b""
        exp = min(want, n_left)
        assert cnt.value == exp and piece == b"".join(recs[off:off + exp])
        if piece:
            p = tmp_path / ("piece_%d" % off)
            assert L.itsx_write_range(os.fsencode(str(p)), ptr, nb.value) == 0, L.itsx_shard_last_error()
            assert p.read_bytes() == piece
        off += exp; n_left -= exp
        assert bool(last.value) == (n_left == 0)
    assert L.itsx_stream_close(h, 0) == 0
    assert L.itsx_write_range(os.fsencode(str(tmp_path / "no" / "such" / "dir")), C.c_char_p(b"x"), 1) != 0
