"""ITSXPRESS_STREAM=1: ONE large FASTQ run as file-order chunks, so that the GPU scores while the file is still being inflated.

The reference reads the whole FASTQ, then runs vsearch, then hmmsearch, then writes (itsxpress/main.py:534-624).  On one MI355X the
hot path of a 10 M-read sample takes 5.5 s and inflating + parsing its .fastq.gz 7.4 s on the box's 16 CPUs: run one after the other
the GPU waits for the loader, then the CPUs for the GPU.  `StreamEngine` overlaps the two behind the interface the mirror classes use:

  * the library's text stream (csrc/fastq_io.cpp TextStream, `itsx_stream_*`) hands out record-aligned slices of the text as the
    block-parallel inflater finishes its rounds;
  * every slice becomes a chunk with a context of its own: a loader thread parses, uploads and dereplicates chunk k + 1 while this
    thread runs the lazy search of chunk k;
  * exactness is the multi-GPU scheme's (itsxpress_amd/multi.py), with ownership decided by FILE ORDER instead of by key:
    a sequence is scored in the chunk where it first occurs (`itsx_keyset_assign`) -- that first occurrence is vsearch's
    representative -- later chunks only point at it; hmmsearch's domZ is summed over the chunks before any threshold is applied
    (bounds after a lazy search; undecided rows settled by counting their profiles in every chunk);
  * everything composed from the chunks' arrays -- per-read coordinates, uc.txt / rep.fa / domtbl.txt -- is `ShardedOps`' code,
    shared with MultiEngine, and equals one Engine's on the whole file (tests/test_gpu_stream.py).

The calls arrive in the reference's order (load, derep, profiles, search); the work is DEFERRED until `search()` knows all of it
and then runs as one pipeline.  Asking for a result earlier (`n_unique`, `get_derep`) runs the load + derep part alone, and
`search()` then goes over the chunks one after the other: same results, no overlap.

The trimmed-FASTQ writer joins the pipeline when the caller says where the output goes BEFORE the search (`plan_output`; class
`_Output` below): thresholds need every chunk's counts, but a chunk finalized right after its own search with PROVISIONAL bounds on
domZ leaves all but a handful of representatives decided for good, and their reads are deflated while the GPU scores the next
chunks.  Without a plan the writer runs after `finalize()` as for every other engine and finds the text in the cache.
"""
import ctypes as C
import os
import queue
import threading

import numpy as np

from . import _lib
from ._lib import EngineError
from .engine import Engine, _device_from_env
from .multi import _HANDLERS, KEY_SEEDS, ShardedOps


def stream_from_env():
    return os.environ.get("ITSXPRESS_STREAM", "").strip() not in ("", "0")


def stream_wanted(path=None, fast=False):
    """ITSXPRESS_STREAM=1: yes; =0: no; unset: yes for a LARGE input in arrays mode (ITSXPRESS_ARRAYS=1), where nothing else reads
    the files between the stages -- ITSX_STREAM_AUTO_MB (512) is the file size from which the pipeline pays (10 M reads: 13.9 s
    staged, 10.5 s streamed, 7.6 s with SeqSample.plan_output as well)"""
    v = os.environ.get("ITSXPRESS_STREAM", "").strip()
    if v != "":
        return v != "0"
    if not fast or not path or not os.path.isfile(path):
        return False
    try:
        return os.path.getsize(path) >= float(os.environ.get("ITSX_STREAM_AUTO_MB", "512") or 512) * (1 << 20)
    except OSError:
        return False


class _TextStream:
    def __init__(self, path, threads=0):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        if threads > 0:
            rc = self.L.itsx_stream_open_threads(os.fsencode(path), int(threads), C.byref(self.h))
        else:
            rc = self.L.itsx_stream_open(os.fsencode(path), C.byref(self.h))
        if rc != 0:
            raise EngineError(rc, self.L.itsx_stream_last_error().decode())

    def next(self, min_bytes):
        ptr, nb, last = C.c_void_p(), C.c_int64(0), C.c_int32(0)
        rc = self.L.itsx_stream_next(self.h, int(min_bytes), C.byref(ptr), C.byref(nb), C.byref(last))
        if rc != 0:
            raise EngineError(rc, self.L.itsx_stream_last_error().decode())
        return ptr.value or 0, nb.value, bool(last.value)

    def next_records(self, n_records):
        """the mate file's slice: exactly n_records records (fewer only when its file ends: the count comes back)"""
        ptr, nb, got, last = C.c_void_p(), C.c_int64(0), C.c_int64(0), C.c_int32(0)
        rc = self.L.itsx_stream_next_records(self.h, int(n_records), C.byref(ptr), C.byref(nb), C.byref(got), C.byref(last))
        if rc != 0:
            raise EngineError(rc, self.L.itsx_stream_last_error().decode())
        return ptr.value or 0, nb.value, got.value, bool(last.value)

    def records_bound(self):
        return int(self.L.itsx_stream_records_bound(self.h)) if self.h else -1

    def close(self, keep=True):
        if self.h:
            h, self.h = self.h, None
            rc = self.L.itsx_stream_close(h, 1 if keep else 0)
            if rc != 0:
                raise EngineError(rc, self.L.itsx_stream_last_error().decode())


class _Output:
    """The trimmed FASTQ written WHILE the chunks are scored (`StreamEngine.plan_output`): right after its own search a chunk is
    finalized with PROVISIONAL bounds on hmmsearch's domZ -- the counts of the chunks so far below, those counts' upper bounds plus
    the reads not yet seen above (the final counts lie in between, so a row both bounds decide alike stays decided) -- and every
    read whose representative has no undecided row goes to the library's writer object at once (`itsx_twriter_*`: units of the
    text sliced and deflated by a pool of threads, written in order).  The few that wait are named after the exact finalize."""

    def __init__(self, L, plan):
        self.L, self.plan = L, plan
        self.w = C.c_void_p()
        rc = L.itsx_twriter_open(os.fsencode(plan["out"]), plan["kind"], 1 if plan["ccs"] else 0, C.byref(self.w))
        if rc != 0:
            raise EngineError(rc, L.itsx_trim_last_error().decode())
        self.g_start = np.full(1 << 20, -1, np.int32)
        self.g_stop = np.full(1 << 20, -1, np.int32)
        self.g_dec = np.ones(1 << 20, np.uint8)
        self.base_ptr = None
        self.reads_seen = 0
        self.late = []                 # per chunk: (first record, indexes of its undecided reads, their global uniques)
        self.n_late_uniques = 0
        self.result = None

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(rc, self.L.itsx_trim_last_error().decode())

    def _grow(self, n):
        if n > self.g_start.shape[0]:
            cap = max(n, 2 * self.g_start.shape[0])
            for name, fill in (("g_start", -1), ("g_stop", -1), ("g_dec", 1)):
                old = getattr(self, name)
                new = np.full(cap, fill, old.dtype)
                new[:old.shape[0]] = old
                setattr(self, name, new)

    def chunk(self, k, eng, st, z_cum, bound, domE):
        """chunk k's search is done and every earlier chunk has been here: thresholds under provisional bounds, rows, reads"""
        plan = self.plan
        z = np.array(z_cum, np.int64)
        half = z.shape[0] // 2
        self.reads_seen += eng.n_reads
        z[half:] += max(0, int(bound) - self.reads_seen)        # every read not yet seen may add one reported target to every profile
        eng.set_partial_coords(True)
        eng.set_domz(z)
        eng.finalize(domE=domE)
        U = eng.n_unique
        gid, v = st["gid"], st["verdict"]
        if U:
            pend = eng.lazy_pending_uniques()
            rows = eng.rep_coords(plan["left"], plan["right"])
            mine = (v[:, 2] == k) & (v[:, 3] == np.arange(U))
            self._grow(int(gid.max()) + 1)
            gm = gid[mine]
            self.g_start[gm] = rows[0][mine]
            self.g_stop[gm] = rows[1][mine]
            self.g_dec[gm] = pend[mine] == 0
            self.n_late_uniques += int((pend[mine] != 0).sum())
        uq = eng.get_derep()[2]
        n = int(uq.shape[0])
        if U:
            g = gid[np.maximum(uq, 0)]
            ok = uq >= 0
            start = np.where(ok, self.g_start[g], -1).astype(np.int32)
            stop = np.where(ok, self.g_stop[g], -1).astype(np.int32)
            dec = np.where(ok, self.g_dec[g], 1).astype(np.uint8)
        else:
            g = np.zeros(n, np.int64)
            start = np.full(n, -1, np.int32); stop = np.full(n, -1, np.int32); dec = np.ones(n, np.uint8)
        self._chk(self.L.itsx_twriter_text(self.w, C.c_void_p(self.base_ptr), int(st["text_end"]), 1 if st["last"] else 0))
        self._chk(self.L.itsx_twriter_coords(self.w, int(st["base"]), n, start.ctypes.data, stop.ctypes.data, dec.ctypes.data))
        idx = np.flatnonzero(dec == 0)
        if idx.shape[0]:
            self.late.append((int(st["base"]), idx.astype(np.int64), g[idx]))

    def settle(self, engs):
        """after the exact finalize: the rows of the representatives that waited, then their reads"""
        plan = self.plan
        if self.n_late_uniques:
            for k, (eng, st) in enumerate(engs):
                U = eng.n_unique
                if not U:
                    continue
                gid, v = st["gid"], st["verdict"]
                mine = (v[:, 2] == k) & (v[:, 3] == np.arange(U))
                waited = mine & (self.g_dec[gid] == 0)
                if waited.any():
                    rows = eng.rep_coords(plan["left"], plan["right"])
                    gw = gid[waited]
                    self.g_start[gw] = rows[0][waited]
                    self.g_stop[gw] = rows[1][waited]
        for base, idx, g in self.late:
            recs = np.ascontiguousarray(base + idx, np.int64)
            a, b = np.ascontiguousarray(self.g_start[g], np.int32), np.ascontiguousarray(self.g_stop[g], np.int32)
            self._chk(self.L.itsx_twriter_update(self.w, recs.ctypes.data, int(recs.shape[0]), a.ctypes.data, b.ctypes.data))
        self.late = []

    def finish(self):
        if self.result is None:
            n, tot = C.c_int64(0), C.c_int64(0)
            w, self.w = self.w, None
            self._chk(self.L.itsx_twriter_close(w, C.byref(n), C.byref(tot)))
            self.result = (n.value, tot.value)
        return self.result

    def abort(self):
        if self.w:
            w, self.w = self.w, None
            self.L.itsx_twriter_close(w, None, None)
            try:
                os.remove(self.plan["out"])
            except OSError:
                pass


class _OutputPaired(_Output):
    """The two mates' trimmed files written while the chunks are scored (`StreamEngine.plan_output_paired`; the reference writes them after
    everything else, itsxpress/SeqSample.py:587-670, 713-790): two of the library's writer objects in slice mode, one per input file, fed
    per chunk with the pair's Python slice bounds -- R1[start:stop] (or [start:] when stop > tlen), R2[tlen - stop : tlen - start] -- for
    every pair whose merged read has both sides (start < stop); a pair that did not merge, or whose read is not trimmed, is skipped in
    both files."""

    SKIP, OPEN = -(1 << 31), (1 << 31) - 1

    def __init__(self, L, plan):
        self.L, self.plan = L, plan
        self.w = self.w2 = None
        ws = []
        try:
            for path in (plan["out"], plan["out2"]):
                w = C.c_void_p()
                rc = L.itsx_twriter_open(os.fsencode(path), plan["kind"], 0, C.byref(w))
                if rc != 0:
                    raise EngineError(rc, L.itsx_trim_last_error().decode())
                ws.append(w)
                self._chk(L.itsx_twriter_set_mode(w, 1))
        except BaseException:
            for w in ws:
                L.itsx_twriter_close(w, None, None)
            raise
        self.w, self.w2 = ws
        self.g_start = np.full(1 << 20, -1, np.int32)
        self.g_stop = np.full(1 << 20, -1, np.int32)
        self.g_dec = np.ones(1 << 20, np.uint8)
        self.g_tlen = np.full(1 << 20, -1, np.int32)
        self.base_ptr = self.base_ptr2 = None
        self.reads_seen = 0
        self.late = []                 # per chunk: (first pair, indexes of its undecided pairs, their global uniques, their tlen)
        self.n_late_uniques = 0
        self.result = None

    def _grow(self, n):
        if n > self.g_start.shape[0]:
            cap = max(n, 2 * self.g_start.shape[0])
            for name, fill in (("g_start", -1), ("g_stop", -1), ("g_dec", 1), ("g_tlen", -1)):
                old = getattr(self, name)
                new = np.full(cap, fill, old.dtype)
                new[:old.shape[0]] = old
                setattr(self, name, new)

    def _pair_bounds(self, start, stop, tlen, ok):
        """the four slice bounds of the pairs whose merged read has (start, stop, tlen); ok: the pair is written"""
        s, e, t = start.astype(np.int64), stop.astype(np.int64), tlen.astype(np.int64)
        keep = ok & (s >= 0) & (e >= 0) & (s < e)
        a1 = np.where(keep, s, 0).astype(np.int32)
        b1 = np.where(keep, np.where(e > t, self.OPEN, e), self.SKIP).astype(np.int32)
        r2s, r2e = t - e, t - s
        a2 = np.where(keep, r2s, 0).astype(np.int32)
        b2 = np.where(keep, np.where(r2e > t, self.OPEN, r2e), self.SKIP).astype(np.int32)
        return a1, b1, a2, b2

    def chunk(self, k, eng, st, z_cum, bound, domE):
        plan = self.plan
        z = np.array(z_cum, np.int64)
        half = z.shape[0] // 2
        self.reads_seen += int(st["pairs"])
        z[half:] += max(0, int(bound) - self.reads_seen)        # every pair not yet seen may add one reported target to every profile
        eng.set_partial_coords(True)
        eng.set_domz(z)
        eng.finalize(domE=domE)
        U = eng.n_unique
        gid, v = st["gid"], st["verdict"]
        if U:
            pend = eng.lazy_pending_uniques()
            rows = eng.rep_coords(plan["left"], plan["right"])
            mine = (v[:, 2] == k) & (v[:, 3] == np.arange(U))
            self._grow(int(gid.max()) + 1)
            gm = gid[mine]
            self.g_start[gm] = rows[0][mine]
            self.g_stop[gm] = rows[1][mine]
            self.g_tlen[gm] = rows[2][mine]                  # (dereplication is exact and full-length: a read is as long as its representative)
            self.g_dec[gm] = pend[mine] == 0
            self.n_late_uniques += int((pend[mine] != 0).sum())
        uq = eng.get_derep()[2]                              # per merged read: its unique (-1: dropped)
        mi = st["merged_index"]                              # per pair: its merged read (-1: not merged)
        npairs = int(mi.shape[0])
        merged = mi >= 0
        if U and npairs and uq.shape[0]:
            u_of_pair = np.where(merged, uq[np.maximum(mi, 0)], -1)
            ok = u_of_pair >= 0
            g = gid[np.maximum(u_of_pair, 0)]
            start = np.where(ok, self.g_start[g], -1)
            stop = np.where(ok, self.g_stop[g], -1)
            dec = np.where(ok, self.g_dec[g], 1).astype(np.uint8)
            tlen = np.where(ok, self.g_tlen[g], 0)
        else:
            g = np.zeros(npairs, np.int64)
            ok = np.zeros(npairs, bool)
            start = np.full(npairs, -1, np.int64); stop = np.full(npairs, -1, np.int64); dec = np.ones(npairs, np.uint8)
            tlen = np.zeros(npairs, np.int64)
        a1, b1, a2, b2 = self._pair_bounds(start, stop, tlen, ok)
        for w, base, end, a, b in ((self.w, self.base_ptr, st["text_end"], a1, b1), (self.w2, self.base_ptr2, st["text_end2"], a2, b2)):
            self._chk(self.L.itsx_twriter_text(w, C.c_void_p(base), int(end), 1 if st["last"] else 0))
            self._chk(self.L.itsx_twriter_coords(w, int(st["pair_base"]), npairs, a.ctypes.data, b.ctypes.data, dec.ctypes.data))
        idx = np.flatnonzero(dec == 0)
        if idx.shape[0]:
            self.late.append((int(st["pair_base"]), idx.astype(np.int64), g[idx]))

    def settle(self, engs):
        plan = self.plan
        if self.n_late_uniques:
            for k, (eng, st) in enumerate(engs):
                U = eng.n_unique
                if not U:
                    continue
                gid, v = st["gid"], st["verdict"]
                mine = (v[:, 2] == k) & (v[:, 3] == np.arange(U))
                waited = mine & (self.g_dec[gid] == 0)
                if waited.any():
                    rows = eng.rep_coords(plan["left"], plan["right"])
                    gw = gid[waited]
                    self.g_start[gw] = rows[0][waited]
                    self.g_stop[gw] = rows[1][waited]
                    self.g_tlen[gw] = rows[2][waited]
        for base, idx, g in self.late:
            recs = np.ascontiguousarray(base + idx, np.int64)
            a1, b1, a2, b2 = self._pair_bounds(self.g_start[g], self.g_stop[g], self.g_tlen[g], np.ones(idx.shape[0], bool))
            self._chk(self.L.itsx_twriter_update(self.w, recs.ctypes.data, int(recs.shape[0]), a1.ctypes.data, b1.ctypes.data))
            self._chk(self.L.itsx_twriter_update(self.w2, recs.ctypes.data, int(recs.shape[0]), a2.ctypes.data, b2.ctypes.data))
        self.late = []

    def finish(self):
        if self.result is None:
            res = []
            for name in ("w", "w2"):
                n, tot = C.c_int64(0), C.c_int64(0)
                w = getattr(self, name)
                setattr(self, name, None)
                self._chk(self.L.itsx_twriter_close(w, C.byref(n), C.byref(tot)))
                res.append((n.value, tot.value))
            self.result = (res[0][0], res[0][1] + res[1][1])
        return self.result

    def abort(self):
        for name, key in (("w", "out"), ("w2", "out2")):
            w = getattr(self, name, None)
            if w:
                setattr(self, name, None)
                self.L.itsx_twriter_close(w, None, None)
                try:
                    os.remove(self.plan[key])
                except OSError:
                    pass


class StreamEngine(ShardedOps):
    """The part of Engine's interface the mirror classes use, over file-order chunks of one FASTQ on one GPU."""

    deferred = True          # the mirror does not ask for counts between its calls (that would split the pipeline)

    def __init__(self, device=None, chunk_mb=None):
        self.L = _lib.lib()
        self.device = _device_from_env() if device is None else int(device)
        if chunk_mb is None:
            chunk_mb = float(os.environ.get("ITSX_STREAM_CHUNK_MB", "0") or 0)
        self.chunk_mb = float(chunk_mb)
        self._engs = []               # [(Engine, state dict)] in file order
        self._path = None
        self._derep_args = (True, 1)
        self._profiles = None
        self._mode = None
        self.rows_mode = -1
        self._search_args = None
        self._loaded = self._searched = self._final = False
        self._n_reads = self._n_unique = 0
        self.n_profiles = 0
        self.n_samples = 1
        self._derep = None
        self._pmeta = None
        self._verdicts, self._bases, self._nloc = [], [], []
        self.timeline = []            # (chunk, seconds since the pipeline started when: text ready, loaded, searched)
        self._plan = None             # plan_output(): the trimmed FASTQ is written while the chunks are scored
        self._out = None
        self._stream = None
        self._path2 = None            # a paired sample (merge_pairs_load): R2, cut at R1's record counts; the chunks' reads are the merged reads
        self._merge = None
        self._stream2 = None
        self._last_merge = None
        self.n_pairs = 0

    # -- plumbing: the handlers of multi.py's workers, called in process
    @property
    def world(self):
        return len(self._engs)

    def _all(self, cmd, *args):
        self._ensure_loaded()                             # (a result asked for before the search: the load + derep part runs now)
        return self._each(cmd, [args] * self.world)

    # after the pipeline every chunk is a small job that leaves most of the GPU idle between its launches (the completion of the
    # undecided profiles above all): the chunks' contexts are independent, so these steps run on a few threads at once
    _CONCURRENT = ("finalize", "complete", "rep_coords", "uniq_of")

    def _each(self, cmd, args_per_shard):
        jobs = list(zip(self._engs, args_per_shard))
        if cmd in self._CONCURRENT and len(jobs) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(int(os.environ.get("ITSX_STREAM_FINISHERS", "8") or 8), len(jobs))) as pool:
                return list(pool.map(lambda j: _HANDLERS[cmd](j[0][0], j[0][1], *j[1]), jobs))
        return [_HANDLERS[cmd](eng, st, *a) for (eng, st), a in jobs]

    def close(self):
        if getattr(self, "_out", None) is not None:
            self._out.abort()                             # (a finished output has no writer left: nothing happens to the file)
            self._out = None
        for eng, _ in getattr(self, "_engs", []):
            eng.close()
        self._engs = []
        if getattr(self, "_plain_eng", None) is not None:  # the plain context of the unstreamed stages (orientation, merging) goes with this object
            try:
                self._plain_eng.close()
            except Exception:
                pass
            self._plain_eng = None
        for name in ("_stream", "_stream2"):
            if getattr(self, name, None) is not None:
                try:
                    getattr(self, name).close(keep=True)
                except Exception:
                    pass
                setattr(self, name, None)

    # -- the output inside the pipeline
    def plan_output(self, outfile, left, right, gzipped=False, zstd_file=False, trim_ccs=False, domE=10.0):
        """Say BEFORE search() where the trimmed reads go (profile-name prefixes of the region's two sides, as ItsPosition has
        them): the writer then works on the chunks that are done while the GPU scores the next ones.  finalize(domE) must use the
        same domE; finish_output() returns (records written, summed length) once finalize() has run."""
        self._plan = {"out": outfile, "left": left, "right": right, "kind": 1 if gzipped else (2 if zstd_file else 0),
                      "ccs": bool(trim_ccs), "domE": float(domE)}

    def plan_output_paired(self, outfile1, outfile2, left, right, gzipped=False, zstd_file=False, domE=10.0):
        """A paired sample (merge_pairs_load): say BEFORE search() where the two mates' trimmed files go; they are then written while the
        chunks are scored (class _OutputPaired)."""
        self._plan = {"out": outfile1, "out2": outfile2, "left": left, "right": right, "kind": 1 if gzipped else (2 if zstd_file else 0),
                      "ccs": False, "domE": float(domE)}

    def output_planned_paired(self, outfile1, outfile2, left, right, gzipped=False, zstd_file=False):
        p = self._plan
        return (self._out is not None and p is not None and p.get("out2") == outfile2 and p["out"] == outfile1 and p["left"] == left
                and p["right"] == right and p["kind"] == (1 if gzipped else (2 if zstd_file else 0)))

    def output_planned(self, outfile, left, right, gzipped=False, zstd_file=False, trim_ccs=False):
        p = self._plan
        return (self._out is not None and p is not None and "out2" not in p and p["out"] == outfile and p["left"] == left and p["right"] == right
                and p["kind"] == (1 if gzipped else (2 if zstd_file else 0)) and p["ccs"] == bool(trim_ccs))

    def finish_output(self):
        if self._out is None or not self._final:
            raise EngineError(-1, "finish_output: no planned output, or finalize() has not run")
        return self._out.finish()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the deferred calls
    def load_reads_file(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.close()
        self._path, self._path2, self._merge, self._last_merge = path, None, None, None
        self._loaded = self._searched = self._final = False
        self._derep = None
        return None

    def merge_pairs_load(self, r1, r2, maxdiffs=40, maxee=2.0, allow_stagger=False):
        """A paired sample, streamed (round 6; the reference merges the whole files first: itsxpress/SeqSample.py:266-365, main.py:513-519):
        R1 and R2 are inflated side by side, R1's slices are cut at record starts and R2's at the same record counts, every pair of slices
        is merged on the device by the chunk's own context (k_merge.hip) and the chunk goes on like a single-end one -- its reads are the
        merged reads.  Deferred like load_reads_file: the counts are known after the pipeline (`n_pairs`, `n_reads`); returns (-1, -1)."""
        for p in (r1, r2):
            if not os.path.exists(p):
                raise FileNotFoundError(p)
        self.close()
        self._path, self._path2 = r1, r2
        self._merge = dict(maxdiffs=int(maxdiffs), maxee=float(maxee), allow_stagger=bool(allow_stagger))
        self._last_merge = dict(r1=r1, r2=r2, **self._merge)
        self._loaded = self._searched = self._final = False
        self._derep = None
        return -1, -1

    def write_merged_fastq(self, path):
        """the merged records as a file after all (a caller that wants the MERGED reads trimmed: one plain context merges once more)"""
        lm = self._last_merge
        if lm is None:
            raise EngineError(-1, "write_merged_fastq: no paired sample was loaded")
        res = self._plain().merge_pairs_files(lm["r1"], lm["r2"], path, maxdiffs=lm["maxdiffs"], maxee=lm["maxee"], allow_stagger=lm["allow_stagger"])
        self._plain_eng.close()
        self._plain_eng = None
        return res

    def derep(self, strand_both=True, minseqlength=1):
        self._derep_args = (bool(strand_both), int(minseqlength))
        if self._loaded:                                  # asked again with other settings: chunk by chunk, as the file was cut
            self._rederep()
            return self._n_unique
        return None

    def cluster(self, cluster_id, strand_both=True):
        if float(cluster_id) >= 1.0:
            return self.derep(strand_both=strand_both, minseqlength=1)     # main.py:534-537: 1.0 is exact dereplication
        raise EngineError(-5, "greedy clustering (cluster_id < 1) is sequential by definition and is not streamed: "
                              "run it without ITSXPRESS_STREAM (DESIGN.md section 7)")

    def load_profiles(self, path=None, text=None):
        if path is not None and not os.path.exists(path):
            raise FileNotFoundError(path)
        self._profiles = (path, text)
        self._final = self._searched = False
        for i, (eng, st) in enumerate(self._engs):
            res = _HANDLERS["profiles"](eng, dict(st, rank=0 if i == 0 else 1), path, text)
            if i == 0:
                self.n_profiles, self._pmeta = int(res[0]), res[1]
        return self.n_profiles if self._engs else None

    @property
    def n_reads(self):
        self._ensure_loaded()
        return self._n_reads

    @n_reads.setter
    def n_reads(self, v):
        self._n_reads = v

    @property
    def n_unique(self):
        self._ensure_loaded()
        return self._n_unique

    @n_unique.setter
    def n_unique(self, v):
        self._n_unique = v

    def _ensure_loaded(self):
        if not self._loaded and self._path is not None:
            self._run_pipeline(with_search=False)

    def _run_pipeline(self, with_search):
        """the pipeline; if the block-parallel inflater gives up on the file AFTER slices have been handed out (a stream it cannot
        split, or one that expands more than 64-fold: the text may not move once slices are out), everything computed so far is
        dropped and the file goes through again with the serial inflater (whole text first, then the same chunks)"""
        try:
            self._pipeline(with_search)
        except EngineError as e:
            if "gave up after slices" not in str(e):
                raise
            keep = os.environ.get("ITSX_PARALLEL_INFLATE")
            os.environ["ITSX_PARALLEL_INFLATE"] = "0"
            try:
                self._pipeline(with_search)
            finally:
                if keep is None:
                    os.environ.pop("ITSX_PARALLEL_INFLATE", None)
                else:
                    os.environ["ITSX_PARALLEL_INFLATE"] = keep

    # -- the pipeline
    def _chunk_bytes(self):
        if self.chunk_mb > 0:
            return int(self.chunk_mb * (1 << 20))
        # slices are 1 - 1.5 x this: about eight chunks of a gzip file (4 x its size is a typical FASTQ), never below 256 MB -- a chunk
        # costs a context and one more round of every launch (10 M reads: 5, 7, 11 chunks took 14.4, 12.1, 13.6 s file to file), and
        # the last chunk's search is the part of the GPU's work nothing overlaps
        fsize = os.path.getsize(self._path)
        return max(256 << 20, int(fsize * 4 / 16))

    def _parse_chunk(self, ptr, nb, base, k, ptr2=None, nb2=0):
        """the loader's first stage: a context, the profiles, the slice's records parsed and packed on the device (a paired sample: both
        slices parsed, the pairs merged on the device, the merged reads packed)"""
        import time
        t0 = time.perf_counter()
        eng = Engine(self.device)
        tc = time.perf_counter()
        st = {"rank": k, "world": 0, "base": int(base)}
        try:
            if self._profiles is not None:
                res = _HANDLERS["profiles"](eng, dict(st, rank=0 if k == 0 else 1), *self._profiles)
                if k == 0:
                    self.n_profiles, self._pmeta = int(res[0]), res[1]
            t1 = time.perf_counter()
            if ptr2 is None:
                eng.load_reads_text(ptr, nb)
            else:
                npairs, _, st["merged_index"] = eng.merge_pairs_load_text(ptr, nb, ptr2, nb2, **self._merge)
                st["pairs"] = int(npairs)
            st["load_s"] = {"context": round(tc - t0, 3), "profiles": round(t1 - tc, 3), "parse+upload": round(time.perf_counter() - t1, 3),
                            "MB": round((nb + nb2) / 1e6, 1), "reads": eng.n_reads}
        except BaseException:
            eng.close()
            raise
        return eng, st

    def _derep_chunk(self, eng, st, k, keyset):
        """the loader's second stage, chunk after chunk in file order: dereplication, and which sequences are seen here first"""
        import time
        t2 = time.perf_counter()
        tup = _HANDLERS["derep"](eng, st, *self._derep_args)
        t3 = time.perf_counter()
        verdict, st["gid"] = self._assign(keyset, tup, k)
        _HANDLERS["verdict"](eng, st, verdict)
        st["load_s"].update({"derep+keys": round(t3 - t2, 3), "ownership": round(time.perf_counter() - t3, 3)})

    def _assign(self, keyset, tup, k):
        U = int(tup.shape[0])
        verdict = np.zeros((max(U, 1), 4), np.int64)
        tup = np.ascontiguousarray(tup, np.int64)
        gid = np.zeros(max(U, 1), np.int64)
        rc = self.L.itsx_keyset_assign(keyset, tup.ctypes.data, U, int(k), verdict.ctypes.data, gid.ctypes.data)
        if rc != 0:
            raise EngineError(rc, self.L.itsx_stream_last_error().decode())
        return verdict[:U], gid[:U]

    def _pipeline(self, with_search):
        import time
        self.close()
        t0 = time.perf_counter()
        q = queue.Queue(maxsize=3)
        keyset = self.L.itsx_keyset_create()
        stop = threading.Event()
        if with_search and self._plan is not None and (("out2" in self._plan) == (self._path2 is not None)):
            self._out = (_OutputPaired if self._path2 is not None else _Output)(self.L, self._plan)
        fin_q, fin_err = queue.Queue(), []

        zs = {}

        def finisher():
            # chunk k is finalized provisionally and handed to the writer when its search and every earlier chunk's are done and the
            # file's record count is known (the inflater is done: a few seconds in) -- on its own thread, beside the later searches
            arrived, nxt = set(), 0
            try:
                while True:
                    item = fin_q.get()
                    if item is not None:
                        arrived.add(item)
                    bound = self._stream.records_bound() if self._stream is not None else -1
                    while nxt in arrived and (bound >= 0 or item is None):
                        zc = np.sum([zs[j] for j in range(nxt + 1)], axis=0)
                        self._out.chunk(nxt, self._engs[nxt][0], self._engs[nxt][1], zc, bound, self._plan["domE"])
                        nxt += 1
                    if item is None:
                        break
            except BaseException as e:                   # noqa: reported by the consumer
                fin_err.append(e)

        fin = None
        if self._out is not None:
            fin = threading.Thread(target=finisher, name="itsx-stream-writer", daemon=True)
            fin.start()

        q1 = queue.Queue(maxsize=2)

        def loader2():
            # second stage on a thread of its own: chunk k is dereplicated (its kernels queue up behind the searches on the GPU)
            # while chunk k + 1 is parsed
            k = 0
            while True:
                item = q1.get()
                if item is None or isinstance(item, BaseException):
                    q.put(item)
                    return
                eng, st, t_text = item
                try:
                    self._derep_chunk(eng, st, k, keyset)
                except BaseException as e:                # noqa: handed to the consumer
                    eng.close()
                    q.put(e)
                    while True:                           # (the first stage may still be filling its queue)
                        it = q1.get()
                        if it is None or isinstance(it, BaseException):
                            return
                        it[0].close()
                q.put((eng, st, t_text, time.perf_counter() - t0))
                k += 1

        def loader():
            stream = None
            try:
                paired = self._path2 is not None
                # (a paired sample's two inflaters share the CPUs: half of the I/O pool's threads each -- ITSX_STREAM_PAIR_THREADS)
                pt = int(os.environ.get("ITSX_STREAM_PAIR_THREADS", "0") or 0) if paired else 0
                stream = _TextStream(self._path, pt)
                self._stream = stream                    # (kept until the engine is closed: the writer reads the text)
                stream2 = None
                if paired:
                    stream2 = _TextStream(self._path2, pt)   # (both files inflate side by side from here on)
                    self._stream2 = stream2
                want = self._chunk_bytes()
                base, k, ptr0, ptr0b, pair_base = 0, 0, None, None, 0
                while not stop.is_set():
                    # (the first slices are smaller -- a quarter, a half -- so that the GPU has something to do early: until the first
                    # chunk is resident nothing overlaps anything)
                    ptr, nb, last = stream.next(max(1, want >> max(0, 2 - k)))    # last: the inflater is done and every member's CRC-32 and length agreed
                    ptr2, nb2 = None, 0
                    if paired:
                        n1 = int(self.L.itsx_count_records(C.c_void_p(ptr), int(nb))) if nb > 0 else 0
                        ptr2, nb2, got, last2 = stream2.next_records(n1)
                        if got != n1 or (last and not last2):
                            # (the reference's merge refuses files of unlike length too: vsearch --fastq_mergepairs stops with an error)
                            raise EngineError(-3, "R1 and R2 hold different numbers of records")
                    t_text = time.perf_counter()
                    if ptr0 is None:
                        ptr0, ptr0b = ptr, ptr2
                        if self._out is not None:
                            self._out.base_ptr = ptr0
                            if paired:
                                self._out.base_ptr2 = ptr0b
                    if nb > 0 or (last and k == 0):
                        eng, st = self._parse_chunk(ptr, nb, base, k, ptr2, nb2)
                        st["text_end"], st["last"] = ptr + nb - ptr0, bool(last)
                        if paired:
                            st["text_end2"], st["pair_base"] = ptr2 + nb2 - ptr0b, pair_base
                            pair_base += st["pairs"]
                        q1.put((eng, st, t_text - t0))
                        base += eng.n_reads
                        k += 1
                    if last:
                        break
                stream = None
                q1.put(None)
            except BaseException as e:                   # noqa: handed on
                q1.put(e)

        th = threading.Thread(target=loader, name="itsx-stream-loader", daemon=True)
        th2 = threading.Thread(target=loader2, name="itsx-stream-derep", daemon=True)
        th.start()
        th2.start()
        err = None
        self.timeline = []
        # two chunks are searched at a time: a lazy search stops a dozen times for a count from the device, and the other context's
        # kernels fill those gaps (the chunks' searches one after the other added up to 6.2-7.0 s per 10 M reads, 5.5 s in one piece)
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("ITSX_STREAM_SEARCHES", "2") or 2)))
        futs = []

        def do_search(k, eng, st, t_text, t_loaded):
            zs[k] = _HANDLERS["search"](eng, st, self._mode, *self._search_args)
            eng.release_scratch()                         # (the next chunk's context takes this one's slab instead of fresh memory)
            self.timeline.append((k, round(t_text, 3), round(t_loaded, 3), round(time.perf_counter() - t0, 3)))
            if fin is not None:
                fin_q.put(k)

        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    err = item
                    break
                eng, st, t_text, t_loaded = item
                self._engs.append((eng, st))
                k = len(self._engs) - 1
                if with_search:
                    futs.append(pool.submit(do_search, k, eng, st, t_text, t_loaded))
                else:
                    self.timeline.append((k, round(t_text, 3), round(t_loaded, 3), round(time.perf_counter() - t0, 3)))
            for f in futs:
                f.result()
        except BaseException as e:                       # noqa
            err = e
        finally:
            pool.shutdown(wait=True)
            self.timeline.sort()
            stop.set()
            while th.is_alive() or th2.is_alive():       # let the loaders get rid of what they still hold
                for qq in (q, q1) if not th2.is_alive() else (q,):
                    try:
                        item = qq.get(timeout=0.05)
                        if isinstance(item, tuple):
                            item[0].close()
                    except queue.Empty:
                        pass
            th.join()
            th2.join()
            self.L.itsx_keyset_destroy(keyset)
            if fin is not None:
                fin_q.put(None)
                fin.join()
                if err is None and fin_err:
                    err = fin_err[0]
        if err is not None:
            self.close()
            raise err
        for i, (_, st) in enumerate(self._engs):
            st["world"] = len(self._engs)
        self._verdicts = [st["verdict"] for _, st in self._engs]
        self._bases = [st["base"] for _, st in self._engs]
        self._nloc = [eng.n_reads for eng, _ in self._engs]
        self._n_reads = int(sum(self._nloc))
        self.n_pairs = int(sum(st.get("pairs", 0) for _, st in self._engs))
        self._index_uniques()
        self._loaded = True
        if with_search:
            self._z = [zs[k] for k in range(len(self._engs))]
            self._searched = True

    def _rederep(self):
        keyset = self.L.itsx_keyset_create()
        try:
            for k, (eng, st) in enumerate(self._engs):
                tup = _HANDLERS["derep"](eng, st, *self._derep_args)
                verdict, st["gid"] = self._assign(keyset, tup, k)
                _HANDLERS["verdict"](eng, st, verdict)
        finally:
            self.L.itsx_keyset_destroy(keyset)
        self._verdicts = [st["verdict"] for _, st in self._engs]
        self._index_uniques()
        self._searched = self._final = False

    # The global unique list (first occurrences in file order) and the chunks' local -> global maps are what the keyset handed out
    # chunk by chunk (`gid`: first-seen order); they are put together when somebody asks -- coordinates per read, the
    # file-compatible outputs --, not in the pipeline: a planned output never needs them.
    def _index_uniques(self):
        self._lazy_index = True
        self._derep = None
        self._final = False
        self._n_unique = 1 + max((int(st["gid"].max()) for _, st in self._engs if st["gid"].shape[0]), default=-1)

    def _build_index(self):
        if not getattr(self, "_lazy_index", False):
            return
        self._lazy_index = False
        n = self._n_unique
        seeds = np.zeros(n, np.int64)
        gm = []
        for k, (_, st) in enumerate(self._engs):
            v, gid = st["verdict"], st["gid"]
            mine = (v[:, 2] == k) & (v[:, 3] == np.arange(v.shape[0]))
            seeds[gid[mine]] = v[mine, 0]
            gm.append(gid)
        if n > 1 and not bool(np.all(seeds[1:] > seeds[:-1])):
            # (a chunk that did not number its uniques by first occurrence: the general route, by sorting)
            keep = (self._derep, self._final)
            ShardedOps._index_uniques(self)
            self._derep, self._final = keep
        else:
            self._seeds_, self._gmap_ = seeds, gm

    @property
    def _seeds(self):
        self._build_index()
        return self._seeds_

    @_seeds.setter
    def _seeds(self, v):
        self._seeds_ = v

    @property
    def _gmap(self):
        self._build_index()
        return self._gmap_

    @_gmap.setter
    def _gmap(self, v):
        self._gmap_ = v

    def finalize(self, domE=10.0):
        if self._out is not None and float(domE) != self._plan["domE"]:
            self._out.abort()                             # planned with another threshold: what was written is void, the caller writes afresh
            self._out = None
        super().finalize(domE)
        if self._out is not None:
            self._out.settle(self._engs)

    # -- a3 / a4
    def search(self, T=10.0, F1=1e-6, F2=1e-6, F3=1e-6):
        if self._profiles is None:
            raise EngineError(-1, "search before load_profiles")
        self._search_args = (T, F1, F2, F3)
        self._final = False
        if not self._loaded:
            if self._path is None:
                raise EngineError(-1, "search before load_reads_file")
            self._run_pipeline(with_search=True)
        else:
            if self._out is not None:                     # (what the writer has was decided by another search: the caller writes afresh)
                self._out.abort()
                self._out = None
            self._z = self._all("search", self._mode, T, F1, F2, F3)
            self._searched = True

    # -- stages that are not streamed: one plain Engine does them
    def _plain(self):
        if getattr(self, "_plain_eng", None) is None:
            self._plain_eng = Engine(self.device)
        return self._plain_eng

    def orient_load_db(self, fasta_path):
        return self._plain().orient_load_db(fasta_path)

    def orient_file(self, fastq):
        res = self._plain().orient_file(fastq)
        # (the whole-file context has done its work: it leaves the device and the host before the streaming run starts -- SeqSample.orient_reads
        # loads the database before every call)
        self._plain_eng.close()
        self._plain_eng = None
        return res

    def merge_pairs_files(self, r1, r2, out, maxdiffs=40, maxee=2.0, allow_stagger=False):
        return self._plain().merge_pairs_files(r1, r2, out, maxdiffs=maxdiffs, maxee=maxee, allow_stagger=allow_stagger)
