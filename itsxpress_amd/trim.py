"""Trimmed-FASTQ writers: the consumers of the coordinates.

Thin wrappers over the native, context-free writers in libitsx_hip.so (trim_host.cpp); they
replace the Biopython loops of Dedup.create_trimmed_seqs / create_paired_trimmed_seqs
(itsxpress/SeqSample.py:713-790, 886-949).  No GPU is needed for these calls.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import EngineError


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def _compression(gzipped, zstd_file):
    """The reference's two flags (SeqSample.py:909-925; gzipped wins) -> the C ABI's compression kind."""
    return 1 if gzipped else (2 if zstd_file else 0)


def read_names(path):
    """labels of a FASTQ file's records, in order, from the native writers' own record parser (itsx_fastq_ids)"""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    L = _lib.lib()
    nb, ob, n = C.c_void_p(), C.c_void_p(), C.c_int64(0)
    rc = L.itsx_fastq_ids(os.fsencode(path), C.byref(nb), C.byref(ob), C.byref(n))
    if rc != 0:
        raise EngineError(rc, L.itsx_trim_last_error().decode())
    try:
        offs = np.ctypeslib.as_array(C.cast(ob, C.POINTER(C.c_int64)), shape=(n.value + 1,)).copy()
        blob = C.string_at(nb, int(offs[-1]))
    finally:
        L.itsx_io_free(nb)
        L.itsx_io_free(ob)
    return [blob[offs[i]:offs[i + 1]].decode() for i in range(n.value)]


def cache_clear():
    """forget the decompressed texts the loaders left for the writers (the library's process-wide cache)"""
    _lib.lib().itsx_io_cache_clear()


def read_text(path):
    """Decompressed bytes of a plain / gzip / zstd file through the engine's reader (gzip.open / pyzstd.open of
    main.py:296-330)."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    L = _lib.lib()
    buf, n = C.c_void_p(), C.c_int64(0)
    rc = L.itsx_io_read(os.fsencode(path), C.byref(buf), C.byref(n))
    if rc != 0:
        raise EngineError(rc, L.itsx_trim_last_error().decode())
    try:
        return C.string_at(buf, n.value)
    finally:
        L.itsx_io_free(buf)


def write_trimmed_fastq(seq_file, outfile, start, stop, gzipped=False, trim_ccs=False, zstd_file=False):
    """record i of seq_file -> record[start[i]:stop[i]] when both >= 0 and start < stop.
    Output plain, gzip (gzipped) or zstd (zstd_file).  Returns (records written, summed trimmed length)."""
    if not os.path.exists(seq_file):
        raise FileNotFoundError(seq_file)
    L = _lib.lib()
    start, stop = _i32(start), _i32(stop)
    n, tot = C.c_int64(0), C.c_int64(0)
    rc = L.itsx_write_trimmed_fastq(os.fsencode(seq_file), os.fsencode(outfile), _compression(gzipped, zstd_file), int(trim_ccs),
                                    start.ctypes.data, stop.ctypes.data, len(start), C.byref(n), C.byref(tot))
    if rc != 0:
        raise EngineError(rc, L.itsx_trim_last_error().decode())
    return n.value, tot.value


def write_trimmed_paired(fastq, fastq2, outfile1, outfile2, names, start, stop, tlen, gzipped=False, trim_ccs=False,
                         zstd_file=False):
    """names[i] = id of merged read i (a list of str, or Engine.read_names_raw()'s (blob, offsets) pair); start/stop/tlen
    per merged read.  Returns pairs written."""
    for p in (fastq, fastq2):
        if not os.path.exists(p):
            raise FileNotFoundError(p)
    # the reference's suffix rule (SeqSample.py:766-787): both .gz, both .zst, or both plain (.fq / .fastq)
    a, b = str(fastq), str(fastq2)
    plain = (".fastq", ".fq")
    if not ((a.endswith(".gz") and b.endswith(".gz")) or (a.endswith(".zst") and b.endswith(".zst")) or
            (a.endswith(plain) and b.endswith(plain))):
        raise ValueError("Fastq and Fastq2 files should both be gzipped (.gz), zstd compressed (.zst) or both be uncompressed. "
                         "Mixed input is not accepted.")
    L = _lib.lib()
    start, stop, tlen = _i32(start), _i32(stop), _i32(tlen)
    if isinstance(names, tuple):
        blob, no = names[0], np.ascontiguousarray(names[1], np.int64)
        n_names = len(no) - 1
    else:
        n_names = len(names)
        no = np.zeros(n_names + 1, np.int64)
        np.cumsum([len(s) for s in names], out=no[1:])
        blob = "".join(names).encode()
    n = C.c_int64(0)
    rc = L.itsx_write_trimmed_paired(os.fsencode(fastq), os.fsencode(fastq2), os.fsencode(outfile1), os.fsencode(outfile2),
                                     _compression(gzipped, zstd_file), int(trim_ccs), C.cast(C.c_char_p(blob), C.c_void_p), no.ctypes.data,
                                     n_names, start.ctypes.data, stop.ctypes.data, tlen.ctypes.data, C.byref(n))
    if rc != 0:
        raise EngineError(rc, L.itsx_trim_last_error().decode())
    return n.value


def coords_from_dicts(names, matchdict, itspos):
    """(start, stop, tlen) arrays over `names` from the reference-shaped dicts (-1 = None / unknown),
    i.e. what Dedup._filterfunc/map_func look up per record (SeqSample.py:814-865)."""
    n = len(names)
    start = np.full(n, -1, np.int32)
    stop = np.full(n, -1, np.int32)
    tlen = np.full(n, -1, np.int32)
    cache = {}
    for i, nm in enumerate(names):
        rep = matchdict.get(nm)
        if rep is None:
            continue
        if rep not in cache:
            try:
                a, b, t = itspos.get_position(rep)
            except KeyError:
                a = b = t = None
            cache[rep] = (-1 if a is None else a, -1 if b is None else b, -1 if t is None else t)
        start[i], stop[i], tlen[i] = cache[rep]
    return start, stop, tlen


def write_oriented_fastq(seq_path, out_path, strand):
    """f4: the FASTQ `vsearch --orient --fastqout` writes for these orientations (+1 as is, -1 reverse-complemented,
    0 dropped).  Returns the number of records written."""
    if not os.path.exists(seq_path):
        raise FileNotFoundError(seq_path)
    st = np.ascontiguousarray(strand, np.int8)
    n = C.c_int64(0)
    L = _lib.lib()
    rc = L.itsx_write_oriented_fastq(os.fsencode(seq_path), os.fsencode(out_path), st.ctypes.data, len(st), C.byref(n))
    if rc != 0:
        raise EngineError(rc, L.itsx_trim_last_error().decode())
    return n.value
