"""ctypes binding of the CPU ORACLE (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the shipped engine (itsxpress_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


class Domain(C.Structure):
    _fields_ = [("seq", C.c_int64), ("prof", C.c_int32), ("tlen", C.c_int32),
                ("ienv", C.c_int32), ("jenv", C.c_int32), ("dom_idx", C.c_int32), ("ndom", C.c_int32),
                ("flags", C.c_int32), ("envsc", C.c_float), ("domcorrection", C.c_float),
                ("dombias", C.c_float), ("bitscore", C.c_float), ("lnP", C.c_double),
                ("seq_score", C.c_float), ("seq_bias", C.c_float),
                ("seq_reported", C.c_int32), ("dom_reported", C.c_int32)]


class PairTrace(C.Structure):
    _fields_ = [("seq", C.c_int64), ("prof", C.c_int32), ("msv_xj", C.c_int32),
                ("pass_msv", C.c_int32), ("pass_bias", C.c_int32), ("pass_fwd", C.c_int32),
                ("msv_sc", C.c_float), ("filtersc", C.c_float), ("fwdsc", C.c_float),
                ("bcksc", C.c_float), ("nullsc", C.c_float), ("nregions", C.c_int32), ("ndom", C.c_int32),
                ("ran_vit", C.c_int32), ("pass_vit", C.c_int32), ("vitsc", C.c_float), ("pad", C.c_int32)]


DOMAIN_DTYPE = np.dtype([("seq", "<i8"), ("prof", "<i4"), ("tlen", "<i4"), ("ienv", "<i4"), ("jenv", "<i4"),
                         ("dom_idx", "<i4"), ("ndom", "<i4"), ("flags", "<i4"), ("envsc", "<f4"),
                         ("domcorrection", "<f4"), ("dombias", "<f4"), ("bitscore", "<f4"), ("lnP", "<f8"),
                         ("seq_score", "<f4"), ("seq_bias", "<f4"), ("seq_reported", "<i4"),
                         ("dom_reported", "<i4")], align=True)
TRACE_DTYPE = np.dtype([("seq", "<i8"), ("prof", "<i4"), ("msv_xj", "<i4"), ("pass_msv", "<i4"),
                        ("pass_bias", "<i4"), ("pass_fwd", "<i4"), ("msv_sc", "<f4"), ("filtersc", "<f4"),
                        ("fwdsc", "<f4"), ("bcksc", "<f4"), ("nullsc", "<f4"), ("nregions", "<i4"),
                        ("ndom", "<i4"), ("ran_vit", "<i4"), ("pass_vit", "<i4"), ("vitsc", "<f4"), ("pad", "<i4")], align=True)


_LIBNAME = "liborc.so"


def build():
    """Compile oracle/liborc.so from the C restatement (gcc only)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, _LIBNAME], check=True)


def use_library(name="liborc.so"):
    """Switch the binding between the CHECKER (liborc.so: plain scalar C, the default, what every parity test uses) and the
    timed CPU BASELINE (libbase_sse.so: the same sources with real SSE2 vectors, bench.py's cpu_baseline leg; results
    bit-identical).  Objects created before the switch must not be used after it."""
    global _LIB, _LIBNAME
    if name != _LIBNAME:
        _LIB, _LIBNAME = None, name


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORACLE_DIR, _LIBNAME)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    assert C.sizeof(Domain) == DOMAIN_DTYPE.itemsize and C.sizeof(PairTrace) == TRACE_DTYPE.itemsize
    L.orc_hmmset_read.restype = C.c_void_p
    L.orc_hmmset_read.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.orc_hmmset_parse.restype = C.c_void_p
    L.orc_hmmset_parse.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int]
    L.orc_hmmset_free.argtypes = [C.c_void_p]
    L.orc_hmmset_count.argtypes = [C.c_void_p]
    L.orc_hmmset_name.restype = C.c_char_p
    L.orc_hmmset_name.argtypes = [C.c_void_p, C.c_int]
    L.orc_hmmset_M.argtypes = [C.c_void_p, C.c_int]
    for f in ("orc_profile_rbv", "orc_profile_rfv", "orc_profile_tfv", "orc_profile_msvparams"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.orc_digitize.argtypes = [C.c_char_p, C.c_int64, C.c_void_p]
    L.orc_search.restype = C.c_void_p
    L.orc_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                             C.c_double, C.c_double, C.c_int, C.c_int]
    L.orc_threshold.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    L.orc_results_free.argtypes = [C.c_void_p]
    L.orc_results_ndom.restype = C.c_int64
    L.orc_results_ndom.argtypes = [C.c_void_p]
    L.orc_results_dom.restype = C.c_void_p
    L.orc_results_dom.argtypes = [C.c_void_p]
    L.orc_results_ntrace.restype = C.c_int64
    L.orc_results_ntrace.argtypes = [C.c_void_p]
    L.orc_results_trace.restype = C.c_void_p
    L.orc_results_trace.argtypes = [C.c_void_p]
    L.orc_results_counts.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_char_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_derep.restype = C.c_int64
    L.orc_derep.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_align_identity.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_align_identity.restype = None
    L.orc_cluster.restype = C.c_int64
    L.orc_cluster.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_merge_pair.restype = C.c_int
    L.orc_merge_pair.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_merge_tables.restype = None
    L.orc_merge_tables.argtypes = [C.c_void_p] * 5
    L.orc_orient_db_add.restype = None
    L.orc_orient_db_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.orc_orient.restype = None
    L.orc_orient.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_xxh64.restype = C.c_uint64
    L.orc_xxh64.argtypes = [C.c_void_p, C.c_int64, C.c_uint64]
    L.orc_det_log.restype = C.c_double
    L.orc_det_log.argtypes = [C.c_double]
    L.orc_det_exp.restype = C.c_double
    L.orc_det_exp.argtypes = [C.c_double]
    L.orc_nullsc.restype = C.c_float
    L.orc_nullsc.argtypes = [C.c_int]
    L.orc_len_lognn3.restype = C.c_double
    L.orc_len_lognn3.argtypes = [C.c_int]
    L.orc_tjb_b.restype = C.c_uint8
    L.orc_tjb_b.argtypes = [C.c_int]
    L.orc_flogsum_table.restype = C.c_void_p
    _LIB = L
    return L


class HmmSet:
    def __init__(self, path=None, text=None):
        L = lib()
        err = C.create_string_buffer(512)
        if path is not None:
            self.h = L.orc_hmmset_read(path.encode(), err, 512)
        else:
            if isinstance(text, str):
                text = text.encode()
            self.h = L.orc_hmmset_parse(text, len(text), err, 512)
        if not self.h:
            raise ValueError(err.value.decode())
        self.n = L.orc_hmmset_count(self.h)
        self.names = [L.orc_hmmset_name(self.h, i).decode() for i in range(self.n)]
        self.M = [L.orc_hmmset_M(self.h, i) for i in range(self.n)]

    def rbv(self, i):
        out = np.zeros(18 * (self.M[i] + 1), np.uint8)
        lib().orc_profile_rbv(self.h, i, out.ctypes.data)
        return out.reshape(18, self.M[i] + 1)

    def Q(self, i):
        return max(2, (self.M[i] + 3) // 4)

    def rfv(self, i):
        out = np.zeros(18 * self.Q(i) * 4, np.float32)
        lib().orc_profile_rfv(self.h, i, out.ctypes.data)
        return out.reshape(18, self.Q(i), 4)

    def tfv(self, i):
        out = np.zeros(8 * self.Q(i) * 4, np.float32)
        lib().orc_profile_tfv(self.h, i, out.ctypes.data)
        return out.reshape(8 * self.Q(i), 4)

    def msvparams(self, i):
        out = np.zeros(4, np.int32)
        lib().orc_profile_msvparams(self.h, i, out.ctypes.data)
        return dict(base=int(out[0]), bias=int(out[1]), tbm=int(out[2]), tec=int(out[3]))

    def __del__(self):
        try:
            lib().orc_hmmset_free(self.h)
        except Exception:
            pass


def digitize(seqs):
    """list[str] -> (codes uint8[total], offsets int64[n+1])"""
    offsets = np.zeros(len(seqs) + 1, np.int64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    codes = np.zeros(max(1, int(offsets[-1])), np.uint8)
    joined = "".join(seqs).encode()
    if lib().orc_digitize(joined, len(joined), codes.ctypes.data) != 0:
        raise ValueError("illegal residue")
    return codes, offsets


class SearchResult:
    def __init__(self, hs, codes, offsets, T=10.0, F1=1e-6, F2=1e-6, F3=1e-6, keep_trace=1, threads=1,
                 domZ=None, domE=10.0):
        L = lib()
        self.hs = hs
        self.nseq = len(offsets) - 1
        self._keep = (codes, offsets)
        self.r = L.orc_search(hs.h, codes.ctypes.data, offsets.ctypes.data, self.nseq, T, F1, F2, F3,
                              keep_trace, threads)
        dz = None
        if domZ is not None:
            dz = np.ascontiguousarray(domZ, np.int64)
        L.orc_threshold(self.r, hs.h, dz.ctypes.data if dz is not None else None, domE)
        n = L.orc_results_ndom(self.r)
        self.domains = np.ctypeslib.as_array(C.cast(L.orc_results_dom(self.r), C.POINTER(C.c_uint8)),
                                             shape=(n * DOMAIN_DTYPE.itemsize,)).view(DOMAIN_DTYPE).copy() \
            if n else np.zeros(0, DOMAIN_DTYPE)
        n = L.orc_results_ntrace(self.r)
        self.trace = np.ctypeslib.as_array(C.cast(L.orc_results_trace(self.r), C.POINTER(C.c_uint8)),
                                           shape=(n * TRACE_DTYPE.itemsize,)).view(TRACE_DTYPE).copy() \
            if n else np.zeros(0, TRACE_DTYPE)
        c = np.zeros(5, np.int64)
        L.orc_results_counts(self.r, c.ctypes.data)
        self.counts = dict(pairs=int(c[0]), past_msv=int(c[1]), past_bias=int(c[2]), past_fwd=int(c[3]),
                           multidomain=int(c[4]))

    def positions(self, left, right):
        start = np.zeros(self.nseq, np.int32)
        stop = np.zeros(self.nseq, np.int32)
        tlen = np.zeros(self.nseq, np.int32)
        ind = np.zeros(self.nseq, np.int32)
        lib().orc_positions(self.r, self.hs.h, self.nseq, left.encode(), right.encode(),
                            start.ctypes.data, stop.ctypes.data, tlen.ctypes.data, ind.ctypes.data)
        return start, stop, tlen, ind

    def __del__(self):
        try:
            lib().orc_results_free(self.r)
        except Exception:
            pass


def derep(codes, offsets, strand_both=True, minlen=1):
    n = len(offsets) - 1
    rep_of = np.zeros(n, np.int64)
    strand = np.zeros(n, np.int8)
    nc = lib().orc_derep(codes.ctypes.data, offsets.ctypes.data, n, int(strand_both), minlen,
                         rep_of.ctypes.data, strand.ctypes.data)
    return int(nc), rep_of, strand


def xxh64(data: bytes, seed=0):
    return int(lib().orc_xxh64(data, len(data), seed))


_MASK4 = np.array([1, 2, 4, 8, 0, 5, 10, 3, 12, 6, 9, 11, 14, 7, 13, 15], np.uint8)


def align_identity(q: str, t: str):
    """(score, matches, counted columns) of the oracle's global alignment (orc_cluster.c)."""
    cq, _ = digitize([q])
    ct, _ = digitize([t])
    mq = np.ascontiguousarray(_MASK4[cq[:len(q)]])
    mt = np.ascontiguousarray(_MASK4[ct[:len(t)]])
    out = (C.c_int64 * 3)()
    lib().orc_align_identity(mq.ctypes.data, len(q), mt.ctypes.data, len(t), C.byref(out, 0), C.byref(out, 8), C.byref(out, 16))
    return int(out[0]), int(out[1]), int(out[2])


def cluster(codes, offsets, labels=None, cluster_id=0.995, strand_both=True, minlen=32):
    """Greedy clustering oracle: returns dict(rep_of, strand, pct_id, order, n_alignments, n_centroids)."""
    n = len(offsets) - 1
    rep_of = np.zeros(n, np.int64)
    strand = np.zeros(n, np.int8)
    pct = np.zeros(n, np.float64)
    order = np.zeros(max(1, n), np.int64)
    stats = np.zeros(2, np.int64)
    lab = loff = None
    if labels is not None:
        enc = [x.encode() for x in labels]
        loff = np.zeros(n + 1, np.int64)
        loff[1:] = np.cumsum([len(x) for x in enc])
        lab = b"".join(enc) + b"\0"
    nk = lib().orc_cluster(codes.ctypes.data, offsets.ctypes.data, n, lab, loff.ctypes.data if loff is not None else None,
                           float(cluster_id), int(strand_both), minlen, rep_of.ctypes.data, strand.ctypes.data,
                           pct.ctypes.data, order.ctypes.data, stats.ctypes.data)
    return dict(rep_of=rep_of, strand=strand, pct_id=pct, order=order[:nk], n_alignments=int(stats[0]), n_centroids=int(stats[1]))


MERGE_REASONS = ["ok", "nokmers", "repeat", "minscore", "maxdiffs", "minovlen", "staggered", "maxee", "empty"]


def merge_pair(f, fq, r, rq, maxdiffs=40, maxee=2.0, allow_stagger=False):
    """vsearch --fastq_mergepairs restated (orc_merge.c): returns (reason, merged_seq, merged_qual, score, shift)."""
    fb, fqb, rb, rqb = f.encode(), fq.encode(), r.encode(), rq.encode()
    out_s = C.create_string_buffer(len(fb) + len(rb) + 1)
    out_q = C.create_string_buffer(len(fb) + len(rb) + 1)
    n = C.c_int(0)
    sc = C.c_double(0)
    sh = C.c_int(0)
    rc = lib().orc_merge_pair(fb, fqb, len(fb), rb, rqb, len(rb), maxdiffs, maxee, int(allow_stagger), out_s, out_q,
                              C.byref(n), C.byref(sc), C.byref(sh))
    return MERGE_REASONS[rc], out_s.raw[:n.value].decode(), out_q.raw[:n.value].decode(), sc.value, sh.value


def merge_tables():
    q2p = np.zeros(128, np.float64)
    match = np.zeros((128, 128), np.float64)
    mism = np.zeros((128, 128), np.float64)
    qsame = np.zeros((128, 128), np.uint8)
    qdiff = np.zeros((128, 128), np.uint8)
    lib().orc_merge_tables(q2p.ctypes.data, match.ctypes.data, mism.ctypes.data, qsame.ctypes.data, qdiff.ctypes.data)
    return q2p, match, mism, qsame, qdiff


def orient(db_seqs, seqs):
    """vsearch --orient restated: returns (strand int8[n] (+1 forward, -1 reverse, 0 undetermined), count_fwd, count_rev)."""
    bits = np.zeros(1 << 21, np.uint8)
    for d in db_seqs:
        c, _ = digitize([d])
        lib().orc_orient_db_add(bits.ctypes.data, c.ctypes.data, len(d))
    codes, off = digitize(seqs)
    n = len(seqs)
    strand = np.zeros(max(1, n), np.int8)
    cf = np.zeros(max(1, n), np.int32)
    cr = np.zeros(max(1, n), np.int32)
    lib().orc_orient(bits.ctypes.data, codes.ctypes.data, off.ctypes.data, n, strand.ctypes.data, cf.ctypes.data, cr.ctypes.data)
    return strand[:n], cf[:n], cr[:n]


def mr_fail_counts(reset=False):
    """multidomain regions that ran into a mirrored bookkeeping limit since the last reset, by kind (index = the engine's
    MrOut.status: 1 unsampleable, 2 > 8 domains in a path, 4 > 512 tuples, 5 walk left the region, 6 > 32 clusters, 7 > 4 envelopes)"""
    out = (C.c_longlong * 8)()
    lib().orc_mr_fail_counts(out, int(reset))
    return [int(x) for x in out]


def dust(seq):
    """vsearch's DUST soft mask of one sequence (orc_dust): a bool array, True = masked"""
    codes, _ = digitize([seq])
    out = np.zeros(max(1, len(seq)), np.uint8)
    lib().orc_dust.restype = None
    lib().orc_dust.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib().orc_dust(codes.ctypes.data, len(seq), out.ctypes.data)
    return out[:len(seq)].astype(bool)
