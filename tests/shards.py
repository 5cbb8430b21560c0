"""Test helper: the multi-rank protocol run by ONE process over several Engine contexts (all on GPU 0), step by step -- exact global
dereplication through the owner's step (itsxpress_amd.multi.owner_verdicts), summed counters, completion of undecided profiles,
coordinates composed from the scorers' rows.  What itsxpress_amd/multi.py does over worker processes and bench.py over RCCL ranks,
without processes: the sizes the GPU suite can afford in one process."""
import ctypes as C

import numpy as np

from itsxpress_amd import multi


def run_shards(engines, shards, hmm_text, left="3_", right="4_", mode="lazy", domE=10.0):
    """engines: one Engine per shard (profiles are loaded here); shards: [(blob, offsets)] in input order.
    Returns (per-read [n, 4] rows of the whole input, summed counters, per-engine stats)."""
    tups, bases = [], []
    base = 0
    for e, (b, o) in zip(engines, shards):
        e.set_rows_mode(mode)
        e.load_profiles(text=hmm_text)
        e.set_reads_buffer(b, o)
        U = e.derep(strand_both=True, minseqlength=1)
        t = np.zeros((max(U, 1), 4), np.int64)
        e._chk(e.L.itsx_unique_keys128(e.h, C.c_uint64(multi.KEY_SEEDS[0]), C.c_uint64(multi.KEY_SEEDS[1]), base, t.ctypes.data))
        tups.append(t[:U])
        bases.append(base)
        base += len(o) - 1
    recv = np.concatenate([np.concatenate([t, np.arange(t.shape[0], dtype=np.int64)[:, None]], axis=1) for t in tups])
    src = np.concatenate([np.full(t.shape[0], r, np.int64) for r, t in enumerate(tups)])
    verdict = multi.owner_verdicts(recv, src)
    cut = np.cumsum([0] + [t.shape[0] for t in tups])
    vs = [verdict[cut[r]:cut[r + 1]] for r in range(len(engines))]
    for r, (e, v) in enumerate(zip(engines, vs)):
        e.set_active_uniques((v[:, 2] == r) & (v[:, 3] == np.arange(v.shape[0])))
        e.search()
    z = np.sum([e.get_domz() for e in engines], axis=0)
    pend, flags = 0, None
    for e in engines:
        e.set_domz(z)
        e.finalize(domE=domE)
        pend = max(pend, e.lazy_pending())
        f = e.lazy_pending_profiles()
        flags = f if flags is None else np.maximum(flags, f)
    if pend > 0:
        for e in engines:
            e.lazy_complete(flags)
        z = np.sum([e.get_domz() for e in engines], axis=0)
        for e in engines:
            e.set_domz(z)
            e.finalize(domE=domE)
            assert e.lazy_pending() == 0
    rep_rows = [np.stack(e.rep_coords(left, right), axis=1) for e in engines]
    out = []
    for r, (e, v) in enumerate(zip(engines, vs)):
        rows = np.empty((v.shape[0], 4), np.int32)
        for s in range(len(engines)):
            m = v[:, 2] == s
            rows[m] = rep_rows[s][v[m, 3]]
        _, _, uq = e.get_derep()
        ok = uq >= 0
        rr = np.full((len(uq), 4), -1, np.int32)
        rr[:, 3] = 0
        rr[ok] = rows[uq[ok]]
        out.append(rr)
    stats = [e.stats() for e in engines]
    for e in engines:
        e.set_rows_mode(None)
    return np.concatenate(out), z, stats
