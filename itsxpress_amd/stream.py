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
`search()` then goes over the chunks one after the other: same results, no overlap.  Not streamed: the trimmed-FASTQ writer (it
needs every threshold, i.e. the last chunk's counts) -- it starts when the last chunk is done and finds the text in the cache.
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


class _TextStream:
    def __init__(self, path):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        rc = self.L.itsx_stream_open(os.fsencode(path), C.byref(self.h))
        if rc != 0:
            raise EngineError(rc, self.L.itsx_stream_last_error().decode())

    def next(self, min_bytes):
        ptr, nb, last = C.c_void_p(), C.c_int64(0), C.c_int32(0)
        rc = self.L.itsx_stream_next(self.h, int(min_bytes), C.byref(ptr), C.byref(nb), C.byref(last))
        if rc != 0:
            raise EngineError(rc, self.L.itsx_stream_last_error().decode())
        return ptr.value or 0, nb.value, bool(last.value)

    def close(self, keep=True):
        if self.h:
            h, self.h = self.h, None
            rc = self.L.itsx_stream_close(h, 1 if keep else 0)
            if rc != 0:
                raise EngineError(rc, self.L.itsx_stream_last_error().decode())


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
            with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
                return list(pool.map(lambda j: _HANDLERS[cmd](j[0][0], j[0][1], *j[1]), jobs))
        return [_HANDLERS[cmd](eng, st, *a) for (eng, st), a in jobs]

    def close(self):
        for eng, _ in getattr(self, "_engs", []):
            eng.close()
        self._engs = []

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
        self._path = path
        self._loaded = self._searched = self._final = False
        self._derep = None
        return None

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
            self._pipeline(with_search=False)

    # -- the pipeline
    def _chunk_bytes(self):
        if self.chunk_mb > 0:
            return int(self.chunk_mb * (1 << 20))
        # slices are 1 - 1.5 x this: about eight chunks of a gzip file (4 x its size is a typical FASTQ), never below 256 MB -- a chunk
        # costs a context and one more round of every launch (10 M reads: 5, 7, 11 chunks took 14.4, 12.1, 13.6 s file to file), and
        # the last chunk's search is the part of the GPU's work nothing overlaps
        fsize = os.path.getsize(self._path)
        return max(256 << 20, int(fsize * 4 / 16))

    def _new_chunk(self, ptr, nb, base, k, keyset):
        import time
        t0 = time.perf_counter()
        eng = Engine(self.device)
        st = {"rank": k, "world": 0, "base": int(base)}
        try:
            if self._profiles is not None:
                res = _HANDLERS["profiles"](eng, dict(st, rank=0 if k == 0 else 1), *self._profiles)
                if k == 0:
                    self.n_profiles, self._pmeta = int(res[0]), res[1]
            t1 = time.perf_counter()
            eng.load_reads_text(ptr, nb)
            t2 = time.perf_counter()
            tup = _HANDLERS["derep"](eng, st, *self._derep_args)
            t3 = time.perf_counter()
            verdict = self._assign(keyset, tup, k)
            _HANDLERS["verdict"](eng, st, verdict)
            t4 = time.perf_counter()
            st["load_s"] = {"context+profiles": round(t1 - t0, 3), "parse+upload": round(t2 - t1, 3), "derep+keys": round(t3 - t2, 3),
                            "ownership": round(t4 - t3, 3), "MB": round(nb / 1e6, 1), "reads": eng.n_reads}
        except BaseException:
            eng.close()
            raise
        return eng, st, time.perf_counter()

    def _assign(self, keyset, tup, k):
        U = int(tup.shape[0])
        verdict = np.zeros((max(U, 1), 4), np.int64)
        tup = np.ascontiguousarray(tup, np.int64)
        rc = self.L.itsx_keyset_assign(keyset, tup.ctypes.data, U, int(k), verdict.ctypes.data)
        if rc != 0:
            raise EngineError(rc, self.L.itsx_stream_last_error().decode())
        return verdict[:U]

    def _pipeline(self, with_search):
        import time
        self.close()
        t0 = time.perf_counter()
        q = queue.Queue(maxsize=3)
        keyset = self.L.itsx_keyset_create()
        stop = threading.Event()

        def loader():
            stream = None
            try:
                stream = _TextStream(self._path)
                want = self._chunk_bytes()
                base, k = 0, 0
                while not stop.is_set():
                    ptr, nb, last = stream.next(want)
                    t_text = time.perf_counter()
                    if nb > 0 or (last and k == 0):
                        eng, st, t_loaded = self._new_chunk(ptr, nb, base, k, keyset)
                        q.put((eng, st, t_text - t0, t_loaded - t0))
                        base += eng.n_reads
                        k += 1
                    if last:
                        break
                stream.close(keep=True)                  # joins the inflater: a corrupt file is reported here at the latest
                stream = None
                q.put(None)
            except BaseException as e:                   # noqa: handed to the consumer
                if stream is not None:
                    try:
                        stream.close(keep=False)
                    except Exception:
                        pass
                q.put(e)

        th = threading.Thread(target=loader, name="itsx-stream-loader", daemon=True)
        th.start()
        err = None
        zs = []
        self.timeline = []
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
                if with_search:
                    zs.append(_HANDLERS["search"](eng, st, self._mode, *self._search_args))
                self.timeline.append((len(self._engs) - 1, round(t_text, 3), round(t_loaded, 3), round(time.perf_counter() - t0, 3)))
        except BaseException as e:                       # noqa
            err = e
        finally:
            stop.set()
            while th.is_alive():                         # let the loader get rid of what it still holds
                try:
                    item = q.get(timeout=0.05)
                    if isinstance(item, tuple):
                        item[0].close()
                except queue.Empty:
                    pass
            th.join()
            self.L.itsx_keyset_destroy(keyset)
        if err is not None:
            self.close()
            raise err
        for i, (_, st) in enumerate(self._engs):
            st["world"] = len(self._engs)
        self._verdicts = [st["verdict"] for _, st in self._engs]
        self._bases = [st["base"] for _, st in self._engs]
        self._nloc = [eng.n_reads for eng, _ in self._engs]
        self._n_reads = int(sum(self._nloc))
        self._index_uniques()
        self._n_unique = int(self._seeds.shape[0])
        self._loaded = True
        if with_search:
            self._z = zs
            self._searched = True

    def _rederep(self):
        keyset = self.L.itsx_keyset_create()
        try:
            for k, (eng, st) in enumerate(self._engs):
                tup = _HANDLERS["derep"](eng, st, *self._derep_args)
                _HANDLERS["verdict"](eng, st, self._assign(keyset, tup, k))
        finally:
            self.L.itsx_keyset_destroy(keyset)
        self._verdicts = [st["verdict"] for _, st in self._engs]
        self._index_uniques()
        self._n_unique = int(self._seeds.shape[0])
        self._searched = self._final = False

    def _index_uniques(self):
        # chunks are in file order and a chunk numbers its uniques by first occurrence: the sequences first seen in chunk k, in that
        # order, ARE the next stretch of the global unique list (checked; anything else takes the general route)
        isnew = [(v[:, 2] == k) & (v[:, 3] == np.arange(v.shape[0])) for k, v in enumerate(self._verdicts)]
        new = [v[m, 0] for v, m in zip(self._verdicts, isnew)]
        seeds = np.concatenate(new) if new else np.zeros(0, np.int64)
        if seeds.shape[0] > 1 and not bool(np.all(seeds[1:] > seeds[:-1])):
            super()._index_uniques()
        else:
            self._seeds = seeds
            self._gmap, at = [], 0
            for v, m in zip(self._verdicts, isnew):
                g = np.empty(v.shape[0], np.int64)
                n = int(m.sum())
                g[m] = at + np.arange(n)
                g[~m] = np.searchsorted(seeds[:at], v[~m, 0])          # first seen in an earlier chunk
                self._gmap.append(g)
                at += n
            self._derep = None
            self._final = False
        self._n_unique = int(self._seeds.shape[0])

    # -- a3 / a4
    def search(self, T=10.0, F1=1e-6, F2=1e-6, F3=1e-6):
        if self._profiles is None:
            raise EngineError(-1, "search before load_profiles")
        self._search_args = (T, F1, F2, F3)
        self._final = False
        if not self._loaded:
            if self._path is None:
                raise EngineError(-1, "search before load_reads_file")
            self._pipeline(with_search=True)
        else:
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
        return self._plain().orient_file(fastq)

    def merge_pairs_files(self, r1, r2, out, maxdiffs=40, maxee=2.0, allow_stagger=False):
        return self._plain().merge_pairs_files(r1, r2, out, maxdiffs=maxdiffs, maxee=maxee, allow_stagger=allow_stagger)
