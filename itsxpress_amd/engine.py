"""Engine: a thin object wrapper over the C ABI (one context = one GPU).

Nothing here computes: every method forwards to libitsx_hip.so and copies results into numpy
arrays the caller owns.  See include/itsx_hip.h for the reference interface each call replaces.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import DOMAIN_DTYPE, STATS_DTYPE, TRACE_DTYPE, EngineError


def _device_from_env():
    for k in ("ITSXPRESS_GPU", "LOCAL_RANK"):
        v = os.environ.get(k)
        if v is not None and v.strip() != "":
            return int(v)
    return 0


class _DevArray:
    """a raw device pointer dressed up for torch.as_tensor (zero-copy): the engine keeps owning the memory"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 3,
                                         "strides": None}


def _devtensor(ptr, shape, typestr, device):
    import torch
    dt = {"<i8": torch.int64, "<i4": torch.int32, "|i1": torch.int8}[typestr]
    if not ptr or int(np.prod(shape)) == 0:
        return torch.empty(tuple(shape), dtype=dt, device=device)
    return torch.as_tensor(_DevArray(ptr, shape, typestr), device=device)


class Engine:
    def __init__(self, device=None):
        self.L = _lib.lib()
        if device is None:
            device = _device_from_env()
        self.h = self.L.itsx_create(int(device), 0)
        if not self.h:
            msg = self.L.itsx_last_error(None).decode()
            raise EngineError(-4, msg)
        self.device = int(device)
        self.n_reads = 0
        self.n_unique = 0
        self.n_profiles = 0
        self.n_samples = 1
        self.rows_mode = -1

    def close(self):
        if getattr(self, "h", None):
            self.L.itsx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(rc, self.L.itsx_last_error(self.h).decode())

    # ---- profiles
    def load_profiles(self, path=None, text=None):
        n = C.c_int(0)
        if path is not None:
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            self._chk(self.L.itsx_load_profiles_file(self.h, os.fsencode(path), C.byref(n)))
        else:
            if isinstance(text, str):
                text = text.encode()
            self._chk(self.L.itsx_load_profiles_mem(self.h, text, len(text), C.byref(n)))
        self.n_profiles = n.value
        return n.value

    def profile_names(self):
        buf = C.create_string_buffer(256)
        out = []
        for i in range(self.n_profiles):
            self._chk(self.L.itsx_profile_name(self.h, i, buf, 256))
            out.append(buf.value.decode())
        return out

    def profile_tables(self, i):
        p = np.zeros(6, np.int32)
        self._chk(self.L.itsx_profile_tables(self.h, i, None, None, None, p.ctypes.data))
        M, Q = int(p[0]), int(p[1])
        rbv = np.zeros((18, M + 1), np.uint8)
        rfv = np.zeros((18, Q, 4), np.float32)
        tfv = np.zeros((8 * Q, 4), np.float32)
        self._chk(self.L.itsx_profile_tables(self.h, i, rbv.ctypes.data, rfv.ctypes.data, tfv.ctypes.data, p.ctypes.data))
        return dict(M=M, Q=Q, base=int(p[2]), bias=int(p[3]), tbm=int(p[4]), tec=int(p[5]), rbv=rbv, rfv=rfv, tfv=tfv)

    # ---- reads
    def set_reads(self, seqs, names=None):
        """seqs: list[str] (or list[bytes]); names: optional list[str]."""
        n = len(seqs)
        lens = np.fromiter((len(s) for s in seqs), np.int64, n)
        offs = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=offs[1:])
        blob = ("".join(seqs)).encode() if (n and isinstance(seqs[0], str)) else b"".join(seqs)
        return self.set_reads_buffer(blob, offs, names)

    def set_reads_buffer(self, blob, offsets, names=None):
        """blob: bytes-like of ASCII bases (bytes, bytearray, numpy uint8 array); the engine keeps a pointer into it
        (itsx_set_reads_view), and this object keeps the buffer alive until the next read set."""
        offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        nb = no = None
        if names is not None:
            nl = np.fromiter((len(s) for s in names), np.int64, n)
            no = np.zeros(n + 1, np.int64)
            np.cumsum(nl, out=no[1:])
            nb = "".join(names).encode()
        if isinstance(blob, np.ndarray):
            blob = np.ascontiguousarray(blob, np.uint8)
            cptr = C.c_void_p(blob.ctypes.data)
        else:
            if not isinstance(blob, bytes):
                blob = bytes(blob)
            cptr = C.cast(C.c_char_p(blob), C.c_void_p)
        self._keep = (blob, offsets, nb, no)
        self._chk(self.L.itsx_set_reads_view(self.h, cptr, offsets.ctypes.data, n,
                                        C.cast(C.c_char_p(nb), C.c_void_p) if nb is not None else None,
                                        no.ctypes.data if no is not None else None))
        self.n_reads = n
        self.n_samples = 1
        return n

    def set_reads_device(self, dev_ptr, offsets, keep=None):
        """ASCII bases already in device memory on this engine's GPU (dev_ptr: integer address); offsets int64[n+1] on the
        host.  `keep`: any object that owns the device buffer (kept referenced until the next read set)."""
        offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        self._keep = (keep, offsets)
        self._chk(self.L.itsx_set_reads_device(self.h, C.c_void_p(int(dev_ptr)), offsets.ctypes.data, n, None, None))
        self.n_reads = n
        self.n_samples = 1
        return n

    def load_reads_file(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        n = C.c_int64(0)
        self._chk(self.L.itsx_load_reads_file(self.h, os.fsencode(path), C.byref(n)))
        self.n_reads = n.value
        self.n_samples = 1
        return n.value

    def load_reads_text(self, ptr, nbytes):
        """the records of FASTA / FASTQ text already in memory (address + length: a slice of itsxpress_amd.stream's text stream)"""
        n = C.c_int64(0)
        self._chk(self.L.itsx_load_reads_text(self.h, C.c_void_p(ptr), int(nbytes), C.byref(n)))
        self.n_reads = n.value
        self.n_samples = 1
        return n.value

    # ---- f4: per-sample batching (many samples, one pass of every kernel, per-sample results)
    def load_reads_files(self, paths):
        """One sequence file per sample, in order: sample index = position in `paths`.
        Returns the read count of each file (reads of sample s are the s-th contiguous block)."""
        for p in paths:
            if not os.path.exists(p):
                raise FileNotFoundError(p)
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        counts = np.zeros(max(1, len(paths)), np.int64)
        self._chk(self.L.itsx_load_reads_files(self.h, arr, len(paths), counts.ctypes.data))
        self.n_reads = int(counts[:len(paths)].sum())
        self.n_samples = max(1, self.L.itsx_num_samples(self.h))
        return counts[:len(paths)]

    def set_samples(self, sample_of_read, n_samples):
        """sample_of_read int32[n_reads] in [0, n_samples); None returns to one sample."""
        if sample_of_read is None:
            self._chk(self.L.itsx_set_samples(self.h, None, 1))
        else:
            a = np.ascontiguousarray(sample_of_read, np.int32)
            assert a.shape[0] == self.n_reads
            self._chk(self.L.itsx_set_samples(self.h, a.ctypes.data if a.size else None, int(n_samples)))
        self.n_samples = max(1, self.L.itsx_num_samples(self.h))

    def select_sample(self, sample):
        """Restrict write_uc / write_rep_fasta / write_domtbl to one sample of the batch (-1 = all)."""
        self._chk(self.L.itsx_select_sample(self.h, int(sample)))

    # ---- derep
    def derep(self, strand_both=True, minseqlength=1):
        n = C.c_int64(0)
        self._chk(self.L.itsx_derep(self.h, int(strand_both), int(minseqlength), C.byref(n)))
        self.n_unique = n.value
        return n.value

    def cluster(self, cluster_id, strand_both=True):
        n = C.c_int64(0)
        self._chk(self.L.itsx_cluster(self.h, float(cluster_id), int(strand_both), C.byref(n)))
        self.n_unique = n.value
        return n.value

    # ---- f4: read orientation
    def orient_load_db(self, fasta_path):
        if not os.path.exists(fasta_path):
            raise FileNotFoundError(fasta_path)
        n = C.c_int64(0)
        self._chk(self.L.itsx_orient_load_db(self.h, os.fsencode(fasta_path), C.byref(n)))
        return n.value

    def orient(self):
        """strand int8[n_reads] (+1 forward, -1 reverse, 0 undetermined), count_fwd, count_rev of the loaded reads"""
        strand = np.zeros(max(1, self.n_reads), np.int8)
        cf = np.zeros(max(1, self.n_reads), np.int32)
        cr = np.zeros(max(1, self.n_reads), np.int32)
        self._chk(self.L.itsx_orient(self.h, strand.ctypes.data, cf.ctypes.data, cr.ctypes.data))
        return strand[:self.n_reads], cf[:self.n_reads], cr[:self.n_reads]

    def orient_file(self, fastq):
        """load a FASTQ file and orient its reads (what SeqSample.orient_reads needs in one call)"""
        self.load_reads_file(fastq)
        return self.orient()

    # ---- f2: paired-end merge
    def merge_pairs(self, fwd, fqual, rev, rqual, maxdiffs=40, maxee=2.0, allow_stagger=False):
        """Lists of equal length (str): forward reads / qualities, reverse reads / qualities as they are in the file.
        Returns (reason int32[n], merged list[(seq, qual) or None], score float64[n], shift int32[n])."""
        n = len(fwd)
        enc = lambda xs: ("".join(xs)).encode()
        fo = np.zeros(n + 1, np.int64); np.cumsum([len(x) for x in fwd], out=fo[1:])
        ro = np.zeros(n + 1, np.int64); np.cumsum([len(x) for x in rev], out=ro[1:])
        fs, fq, rs, rq = enc(fwd), enc(fqual), enc(rev), enc(rqual)
        assert len(fs) == len(fq) and len(rs) == len(rq)
        cap = int(fo[-1] + ro[-1]) + 1
        oseq = C.create_string_buffer(cap)
        oqual = C.create_string_buffer(cap)
        olen = np.zeros(max(1, n), np.int32)
        reason = np.zeros(max(1, n), np.int32)
        score = np.zeros(max(1, n), np.float64)
        shift = np.zeros(max(1, n), np.int32)
        self._chk(self.L.itsx_merge_buffers(self.h, fs, fq, fo.ctypes.data, rs, rq, ro.ctypes.data, n, int(maxdiffs), float(maxee),
                                            int(allow_stagger), oseq, oqual, olen.ctypes.data, reason.ctypes.data,
                                            score.ctypes.data, shift.ctypes.data))
        merged = []
        for i in range(n):
            o = int(fo[i] + ro[i])
            merged.append((oseq.raw[o:o + olen[i]].decode(), oqual.raw[o:o + olen[i]].decode()) if reason[i] == 0 else None)
        return reason[:n], merged, score[:n], shift[:n]

    def merge_pairs_files(self, r1, r2, out, maxdiffs=40, maxee=2.0, allow_stagger=False):
        for p in (r1, r2):
            if not os.path.exists(p):
                raise FileNotFoundError(p)
        n = C.c_int64(0)
        m = C.c_int64(0)
        self._chk(self.L.itsx_merge_pairs_files(self.h, os.fsencode(r1), os.fsencode(r2), os.fsencode(out), int(maxdiffs),
                                                float(maxee), int(allow_stagger), C.byref(n), C.byref(m)))
        return n.value, m.value

    def merge_pairs_load(self, r1, r2, maxdiffs=40, maxee=2.0, allow_stagger=False):
        """merge R1 / R2 and leave the merged reads as this engine's read set (nothing written): (pairs, merged)"""
        for p in (r1, r2):
            if not os.path.exists(p):
                raise FileNotFoundError(p)
        n = C.c_int64(0)
        m = C.c_int64(0)
        self._chk(self.L.itsx_merge_pairs_load(self.h, os.fsencode(r1), os.fsencode(r2), int(maxdiffs), float(maxee), int(allow_stagger),
                                               C.byref(n), C.byref(m)))
        self.n_reads, self.n_samples, self.n_unique = m.value, 1, 0
        self._last_merge = dict(r1=r1, r2=r2, maxdiffs=int(maxdiffs), maxee=float(maxee), allow_stagger=bool(allow_stagger))
        return n.value, m.value

    def merge_pairs_load_text(self, ptr1, nb1, ptr2, nb2, maxdiffs=40, maxee=2.0, allow_stagger=False):
        """the same from record-aligned pieces of R1's and R2's text in memory that hold the same number of records (addresses + lengths:
        slices of itsxpress_amd.stream's two text streams): (pairs, merged, per pair the index of its merged read or -1)"""
        n = C.c_int64(0)
        m = C.c_int64(0)
        self._chk(self.L.itsx_merge_pairs_load_text(self.h, C.c_void_p(ptr1), int(nb1), C.c_void_p(ptr2), int(nb2), int(maxdiffs), float(maxee),
                                                    int(allow_stagger), C.byref(n), C.byref(m)))
        self.n_reads, self.n_samples, self.n_unique = m.value, 1, 0
        idx = np.zeros(max(1, n.value), np.int32)
        self._chk(self.L.itsx_merge_pair_index(self.h, idx.ctypes.data, n.value))
        return n.value, m.value, idx[:n.value]

    def write_merged_fastq(self, path):
        """After merge_pairs_load (which writes nothing): the merged records as a FASTQ file after all -- the same merge once more, in a
        context of its own (this engine's read set and results stay as they are); record i of the file is read i of this engine.  For
        the consumer that wants the MERGED reads trimmed (Dedup.create_trimmed_seqs on a paired sample, itsxpress/main.py:596-624)."""
        lm = getattr(self, "_last_merge", None)
        if lm is None:
            raise EngineError(-1, "write_merged_fastq: this engine's reads do not come from merge_pairs_load")
        tmp = Engine(self.device)
        try:
            n, m = tmp.merge_pairs_files(lm["r1"], lm["r2"], path, maxdiffs=lm["maxdiffs"], maxee=lm["maxee"], allow_stagger=lm["allow_stagger"])
        finally:
            tmp.close()
        if m != self.n_reads:
            raise EngineError(-1, "write_merged_fastq: %d merged records written, the engine holds %d" % (m, self.n_reads))
        return m

    def get_cluster(self):
        """After cluster(id < 1): (pct_id float64[n_reads] (-1 for centroids / dropped), order int64[kept])."""
        pct = np.zeros(self.n_reads, np.float64)
        order = np.zeros(max(1, self.n_reads), np.int64)
        n = C.c_int64(0)
        self._chk(self.L.itsx_get_cluster(self.h, pct.ctypes.data, order.ctypes.data, C.byref(n)))
        return pct, order[:n.value]

    def get_derep(self):
        rep_of = np.zeros(self.n_reads, np.int64)
        strand = np.zeros(self.n_reads, np.int8)
        uniq_of = np.zeros(self.n_reads, np.int64)
        self._chk(self.L.itsx_get_derep(self.h, rep_of.ctypes.data, strand.ctypes.data, uniq_of.ctypes.data))
        return rep_of, strand, uniq_of

    def unique_keys(self, seed=0):
        """XXH64(seed) of every local unique's forward strand and reverse complement (uint64[U] each)."""
        kf = np.zeros(max(1, self.n_unique), np.uint64)
        kr = np.zeros(max(1, self.n_unique), np.uint64)
        self._chk(self.L.itsx_unique_keys(self.h, C.c_uint64(seed), kf.ctypes.data, kr.ctypes.data))
        return kf[:self.n_unique], kr[:self.n_unique]

    def set_active_uniques(self, active):
        a = np.ascontiguousarray(active, np.uint8)
        assert a.shape[0] == self.n_unique
        self._chk(self.L.itsx_set_active_uniques(self.h, a.ctypes.data if a.size else None))

    def read_names_raw(self):
        """labels of the loaded reads, in input order: (bytes blob, int64 offsets[n_reads + 1]) -- the form the paired
        writer takes without a detour through a million Python strings"""
        offs = np.zeros(self.n_reads + 1, np.int64)
        self._chk(self.L.itsx_get_read_names(self.h, None, 0, offs.ctypes.data))
        buf = C.create_string_buffer(int(offs[-1]) + 1)
        self._chk(self.L.itsx_get_read_names(self.h, buf, int(offs[-1]), offs.ctypes.data))
        return buf.raw[:int(offs[-1])], offs

    def read_names(self):
        """labels of the loaded reads, in input order (list[str])"""
        blob, offs = self.read_names_raw()
        blob = blob.decode()
        return [blob[offs[i]:offs[i + 1]] for i in range(self.n_reads)]

    def get_uniques(self):
        seed = np.zeros(self.n_unique, np.int64)
        ab = np.zeros(self.n_unique, np.int64)
        self._chk(self.L.itsx_get_uniques(self.h, seed.ctypes.data, ab.ctypes.data))
        return seed, ab

    # ---- search
    def search(self, T=10.0, F1=1e-6, F2=1e-6, F3=1e-6):
        self._chk(self.L.itsx_search(self.h, T, F1, F2, F3))

    ROWS_FULL, ROWS_COMPACT, ROWS_LAZY = 0, 1, 2

    def set_rows_mode(self, mode):
        """What search() keeps of the domain table: "full" (every row: domtbl.txt), "compact" (every pair evaluated, only the rows
        that can still win ItsPosition's argmax kept), "lazy" (pairs that cannot win it are not evaluated past their Forward score;
        same coordinates), None = from the environment (ITSX_ROWS / ITSX_COMPACT_ROWS)."""
        m = {None: -1, "env": -1, "full": 0, "compact": 1, "lazy": 2}.get(mode, mode)
        self._chk(self.L.itsx_set_rows_mode(self.h, int(m)))
        self.rows_mode = int(m)

    def set_kept_rows(self, on=True):
        """After a "compact" / "lazy" search: domains() / write_domtbl() serve the WINNERS among the rows the context kept (per target and
        2-character profile prefix the reported row ItsPosition's argmax ends up with -- a row of the full table) instead of refusing."""
        self._chk(self.L.itsx_set_kept_rows(self.h, 1 if on else 0))

    def lazy_pending(self):
        """after finalize() of a lazy search whose counters were exchanged: rows that still depend on the exact domZ (> 0: search
        again in "compact" mode on every rank)"""
        return int(self.L.itsx_lazy_pending(self.h))

    def lazy_pending_profiles(self):
        """int32[n_profiles]: 1 where a pending row belongs to the profile"""
        f = np.zeros(max(1, self.n_profiles), np.int32)
        self._chk(self.L.itsx_lazy_pending_profiles(self.h, f.ctypes.data))
        return f[:self.n_profiles]

    def lazy_pending_uniques(self):
        """uint8[n_unique]: 1 where an undecided row could still change the representative's coordinates"""
        f = np.zeros(max(1, self.n_unique), np.uint8)
        self._chk(self.L.itsx_lazy_pending_uniques(self.h, f.ctypes.data))
        return f[:self.n_unique]

    def set_partial_coords(self, on=True):
        """rep_coords / trim_coords answer although rows are undecided (the caller skips the flagged representatives)"""
        self._chk(self.L.itsx_set_partial_coords(self.h, 1 if on else 0))

    def lazy_complete(self, flags):
        """count the flagged profiles' reported targets exactly (every pair of theirs is evaluated); then exchange / finalize again"""
        f = np.ascontiguousarray(flags, np.int32)
        assert f.size == self.n_profiles
        self._chk(self.L.itsx_lazy_complete(self.h, f.ctypes.data))

    def get_domz(self):
        z = np.zeros(max(1, int(self.L.itsx_domz_count(self.h))), np.int64)      # [sample][profile] (x 2 after a lazy search: lower, upper bounds)
        self._chk(self.L.itsx_get_domz(self.h, z.ctypes.data))
        return z[:int(self.L.itsx_domz_count(self.h))]

    def set_domz(self, z):
        z = np.ascontiguousarray(z, np.int64)
        assert z.size == int(self.L.itsx_domz_count(self.h))
        self._chk(self.L.itsx_set_domz(self.h, z.ctypes.data))

    # ---- device-resident exchange (multi-GPU drivers): torch tensors over the engine's own device memory, no copies
    def _torch_device(self, device=None):
        import torch
        return device if device is not None else torch.device("cuda", self.device)

    def domz_device(self, device=None):
        """int64[n_samples * n_profiles] reported-target counters ON THE DEVICE: all-reduce in place, then finalize()."""
        p, n = C.c_void_p(0), C.c_int64(0)
        self._chk(self.L.itsx_domz_device(self.h, C.byref(p), C.byref(n)))
        return _devtensor(p.value, (n.value,), "<i8", self._torch_device(device))

    def _coords_device(self, fn, left, right, device):
        p, n = C.c_void_p(0), C.c_int64(0)
        self._chk(fn(self.h, left.encode(), right.encode(), C.byref(p), C.byref(n)))
        return _devtensor(p.value, (n.value, 4), "<i4", self._torch_device(device))

    def trim_coords_device(self, left, right, device=None):
        """int32 [n_reads, 4] rows (start, stop, tlen, in_ddict) on the device."""
        return self._coords_device(self.L.itsx_trim_coords_device, left, right, device)

    def rep_coords_device(self, left, right, device=None):
        return self._coords_device(self.L.itsx_rep_coords_device, left, right, device)

    def derep_device(self, device=None):
        """dict of device tensors: rep_of, uniq_of (int32[n_reads]), strand (int8[n_reads]), seed_read (int32[n_unique])."""
        a, b, c, d = C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0)
        self._chk(self.L.itsx_derep_device(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        dev = self._torch_device(device)
        return dict(rep_of=_devtensor(a.value, (self.n_reads,), "<i4", dev), uniq_of=_devtensor(b.value, (self.n_reads,), "<i4", dev),
                    strand=_devtensor(c.value, (self.n_reads,), "|i1", dev), seed_read=_devtensor(d.value, (self.n_unique,), "<i4", dev))

    def unique_tuples(self, gidx_base, device=None, seeds=(0x1F83D9ABFB41BD6B, 0x5BE0CD19137E2179)):
        """int64 [n_unique, 4] rows on the device: the orientation-free 128-bit key of each local unique (two XXH64 seeds),
        gidx_base + index of its first occurrence, 1 if the forward strand is the canonical orientation."""
        p, n = C.c_void_p(0), C.c_int64(0)
        self._chk(self.L.itsx_unique_keys128_device(self.h, C.c_uint64(seeds[0]), C.c_uint64(seeds[1]), int(gidx_base), C.byref(p), C.byref(n)))
        return _devtensor(p.value, (n.value, 4), "<i8", self._torch_device(device))

    def finalize(self, domE=10.0):
        self._chk(self.L.itsx_search_finalize(self.h, domE))

    def domains(self):
        n = self.L.itsx_num_domains(self.h)
        if n < 0:
            raise EngineError(-1, self.L.itsx_last_error(self.h).decode())
        out = np.zeros(n, DOMAIN_DTYPE)
        if n:
            self._chk(self.L.itsx_get_domains(self.h, out.ctypes.data))
        return out

    def pairtraces(self):
        n = self.L.itsx_num_pairtraces(self.h)
        if n < 0:
            raise EngineError(-1, self.L.itsx_last_error(self.h).decode())
        out = np.zeros(n, TRACE_DTYPE)
        if n:
            self._chk(self.L.itsx_get_pairtraces(self.h, out.ctypes.data, TRACE_DTYPE.itemsize))
        return out

    def _coords(self, fn, n, left, right):
        a = [np.zeros(max(n, 1), np.int32) for _ in range(4)]
        self._chk(fn(self.h, left.encode(), right.encode(), *[x.ctypes.data for x in a]))
        return tuple(x[:n] for x in a)

    def trim_coords(self, left, right):
        """per READ: start, stop, tlen (-1 = None), in_ddict."""
        return self._coords(self.L.itsx_trim_coords, self.n_reads, left, right)

    def rep_coords(self, left, right):
        return self._coords(self.L.itsx_rep_coords, self.n_unique, left, right)

    # ---- writers
    def write_uc(self, path):
        self._chk(self.L.itsx_write_uc(self.h, os.fsencode(path)))

    def write_rep_fasta(self, path):
        self._chk(self.L.itsx_write_rep_fasta(self.h, os.fsencode(path)))

    def write_domtbl(self, path):
        self._chk(self.L.itsx_write_domtbl(self.h, os.fsencode(path)))

    def stats(self):
        s = np.zeros(1, STATS_DTYPE)
        self._chk(self.L.itsx_get_stats(self.h, s.ctypes.data, STATS_DTYPE.itemsize))
        return {k: (s[0][k].item() if s[0][k].ndim == 0 else s[0][k].tolist()) for k in STATS_DTYPE.names}

    def release_scratch(self):
        """hand the context's DP slab (tens of GB) to the next context of this process on the device: a streamed file's chunk after
        its search (results and what finalize / complete need stay)"""
        self._chk(self.L.itsx_release_scratch(self.h))

    def switches(self, now=False):
        """the library's environment switches that were set when the last search started (now=True: that are set now), as a dict
        NAME -> value; a test hook that is not honoured (no ITSX_TEST_HOOKS=1) carries the note "(ignored: ...)" """
        h = None if now else self.h
        n = self.L.itsx_switches(h, None, 0)
        buf = C.create_string_buffer(int(n))
        self.L.itsx_switches(h, buf, n)
        return dict(line.split("=", 1) for line in buf.value.decode().split("\n") if "=" in line)

    # ---- test hooks
    def debug_read_hashes(self):
        f = np.zeros(self.n_reads, np.uint64)
        r = np.zeros(self.n_reads, np.uint64)
        self._chk(self.L.itsx_debug_read_hashes(self.h, f.ctypes.data, r.ctypes.data))
        return f, r

    def debug_packed_read(self, i):
        nw, ne = C.c_int32(0), C.c_int32(0)
        self._chk(self.L.itsx_debug_packed_read(self.h, i, None, C.byref(nw), None, C.byref(ne)))
        w = np.zeros(max(nw.value, 1), np.uint32)
        e = np.zeros(max(ne.value, 1), np.uint32)
        self._chk(self.L.itsx_debug_packed_read(self.h, i, w.ctypes.data, C.byref(nw), e.ctypes.data, C.byref(ne)))
        return w[:nw.value], e[:ne.value]

    def debug_dust(self, lengths):
        """the device's DUST soft mask of the current reads: a list of bool arrays"""
        tot = int(sum(lengths))
        out = np.zeros(max(tot, 1), np.uint8)
        self._chk(self.L.itsx_debug_dust(self.h, out.ctypes.data))
        o = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        return [out[o[i]:o[i + 1]].astype(bool) for i in range(len(lengths))]

    def debug_logf(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros_like(x)
        self._chk(self.L.itsx_debug_logf(self.h, x.ctypes.data, len(x), out.ctypes.data))
        return out

    def debug_detmath(self, x):
        x = np.ascontiguousarray(x, np.float64)
        a = np.zeros_like(x)
        b = np.zeros_like(x)
        self._chk(self.L.itsx_debug_detmath(self.h, x.ctypes.data, len(x), a.ctypes.data, b.ctypes.data))
        return a, b


def read_fastx(path):
    """Minimal FASTA/FASTQ reader (plain, .gz or .zst, through the engine's file reader) used by the host side,
    tests and the bench: returns (names, seqs)."""
    from .trim import read_text
    lines = read_text(path).decode().split("\n")
    names, seqs = [], []
    if not lines or not lines[0]:
        return names, seqs
    if lines[0][0] == "@":
        i = 0
        while i < len(lines) and lines[i]:
            names.append(lines[i][1:].split()[0])
            seqs.append(lines[i + 1].rstrip("\r") if i + 1 < len(lines) else "")
            i += 4
    else:
        cur = []
        for line in lines:
            if line[:1] == ">":
                if names:
                    seqs.append("".join(cur))
                cur = []
                names.append(line[1:].split()[0])
            else:
                cur.append(line.strip())
        seqs.append("".join(cur))
    return names, seqs
