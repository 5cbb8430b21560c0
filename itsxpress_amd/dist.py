"""Multi-GPU glue for the path: one process per GPU, reads sharded by contiguous index range.

The path has exactly two exchange steps, both tiny:
  * all-reduce(sum) of the per-profile reported-target counts -- hmmsearch's domZ, which the
    domain E-value threshold needs for the WHOLE data set (int64[P], <= ~12 KB);
  * gather of the per-read trim coordinates to rank 0 (int32 x 4 per read).
torch.distributed is used as plumbing only (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).
"""
import numpy as np


def shard_bounds(n_items, world_size, rank):
    """contiguous shard [lo, hi) of rank; input order is kept inside a shard (first-occurrence
    representative choice depends on it)."""
    lo = n_items * rank // world_size
    hi = n_items * (rank + 1) // world_size
    return lo, hi


def assign_samples(reads_per_sample, world_size):
    """Whole samples to ranks for a sample batch (itsxpress_amd/batch.py): samples are independent runs of the path
    (dereplication and domZ are per sample), so a multi-GPU batch needs no exchange at all -- every rank batches its own
    samples.  Longest-processing-time assignment on the read counts; returns one ascending list of sample indices per rank."""
    order = sorted(range(len(reads_per_sample)), key=lambda i: (-int(reads_per_sample[i]), i))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(reads_per_sample[i])
    return [sorted(x) for x in out]


def allreduce_domz(domz, device=None):
    """sum the per-profile counts over all ranks; returns int64 numpy array."""
    import torch
    import torch.distributed as dist
    z = np.ascontiguousarray(domz, np.int64)
    if not (dist.is_available() and dist.is_initialized()):
        return z
    t = torch.from_numpy(z.copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


_PINNED = {}


def _pinned(shape, dtype):
    """a cached page-locked host tensor (device -> host copies of the gathered block run at PCIe speed, not at the
    pageable-copy rate); plain memory when no GPU is in use (gloo tests)"""
    import torch
    key = (tuple(shape), dtype)
    t = _PINNED.get(key)
    if t is None:
        try:
            t = torch.empty(shape, dtype=dtype, pin_memory=torch.cuda.is_available())
        except RuntimeError:
            t = torch.empty(shape, dtype=dtype)
        _PINNED.clear()                                  # one shape at a time: a bench step always asks for the same one
        _PINNED[key] = t
    return t


def gather_coords(start, stop, tlen, in_ddict, device=None, dst=0):
    """rank dst receives every rank's [n_i, 4] int32 block (shards may differ in size); the others get None.
    The final gather of the path: 16 B per read, sent to dst only."""
    import torch
    import torch.distributed as dist
    block = np.stack([start, stop, tlen, in_ddict], axis=1).astype(np.int32)
    if not (dist.is_available() and dist.is_initialized()):
        return [block]
    ws, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([block.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(x) for x in torch.cat(sizes).cpu().tolist()]
    nmax = max(max(sizes), 1)
    if block.shape[0] == nmax:
        pad = block
    else:
        pad = np.full((nmax, 4), -1, np.int32)
        pad[:block.shape[0]] = block
    t = torch.from_numpy(np.ascontiguousarray(pad))
    if device is not None:
        t = t.to(device)
    if rank == dst:
        big = torch.empty((ws, nmax, 4), dtype=torch.int32, device=t.device)
        dist.gather(t, [big[r] for r in range(ws)], dst=dst)
        host = _pinned((ws, nmax, 4), torch.int32) if t.is_cuda else None
        if host is not None:
            host.copy_(big)
            arr = host.numpy()
        else:
            arr = big.numpy()
        return [arr[r, :sizes[r]].copy() for r in range(ws)]
    dist.gather(t, None, dst=dst)
    return None


def gather_rows(rows, dst=0):
    """rows: [n_i, 4] int32 tensor per rank, WHEREVER it lives (the engine's device memory on the GPU box, host memory
    under gloo).  Rank dst receives every rank's block as numpy arrays, the others None: the final gather of the path
    (SURVEY 8e), 16 B per read over RCCL, straight out of the buffer the engine wrote."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [rows.cpu().numpy()]
    if rows.is_cuda and _host_backend():
        rows = rows.cpu()
    ws, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(x) for x in torch.cat(sizes).cpu().tolist()]
    nmax = max(max(sizes), 1)
    if rows.shape[0] == nmax:
        pad = rows.contiguous()
    else:
        pad = torch.full((nmax, 4), -1, dtype=torch.int32, device=rows.device)
        pad[:rows.shape[0]] = rows
    if rank == dst:
        big = torch.empty((ws, nmax, 4), dtype=torch.int32, device=rows.device)
        dist.gather(pad, [big[r] for r in range(ws)], dst=dst)
        if big.is_cuda:
            host = _pinned((ws, nmax, 4), torch.int32)
            host.copy_(big)
            arr = host.numpy()
        else:
            arr = big.numpy()
        return [arr[r, :sizes[r]].copy() for r in range(ws)]
    dist.gather(pad, None, dst=dst)
    return None


def _host_backend():
    """gloo moves host memory: device tensors take a detour through the host there (tests); RCCL takes them as they are"""
    import torch.distributed as dist
    return dist.get_backend() == "gloo"


def allreduce_domz_device(engine, device=None):
    """sum hmmsearch's domZ over all ranks where the counters live: the engine's device buffer is all-reduced in place
    (RCCL) and itsx_search_finalize then reads it there."""
    import torch.distributed as dist
    t = engine.domz_device(device)
    if dist.is_available() and dist.is_initialized() and t.numel():
        if t.is_cuda and _host_backend():
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if t.is_cuda:
            # itsx_search_finalize reads these counters on the ENGINE's stream, which knows nothing of torch's: the
            # reduction (and the copy above) must have landed before this returns, whatever stream torch is using
            # (a torch.cuda.stream(...) context, a per-thread default stream).  The zero-copy *_device views all need
            # this ordering: torch work on engine memory is synchronised before the engine touches it again.
            import torch
            torch.cuda.current_stream(t.device).synchronize()
    return t


def agree(ok, device=None):
    """Every rank learns whether ALL ranks are fine before the next collective: a rank whose engine call failed must not leave its
    peers blocked in an all-reduce it never joins.  Returns True when every rank passed ok=True."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return bool(ok)
    t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=None if (device is None or _host_backend()) else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item()) == 0


def exchange_and_finalize(engine, device=None, domE=10.0, search=None):
    """The step between itsx_search and the coordinates on N ranks: all-reduce the counters where they live, apply the thresholds.
    After a LAZY search (csrc/k_lazy.hip) the counters are bounds on hmmsearch's domZ; a rank may then hold rows that neither bound
    decides and that could change a coordinate (engine.lazy_pending()).  The ranks agree on the maximum; if it is positive the
    profiles of those rows (OR over the ranks) are counted exactly on EVERY rank (engine.lazy_complete: every pair of theirs is
    evaluated), the counters are exchanged again and the thresholds applied again.  `search` (the callable that runs
    engine.search with the caller's thresholds) is the safety net should rows stay undecided even then.  Returns the number of
    pending rows before the completion (0 = nothing had to be completed)."""
    import torch
    import torch.distributed as dist
    allreduce_domz_device(engine, device)
    engine.finalize(domE=domE)
    pend = engine.lazy_pending()
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([pend], dtype=torch.int64, device=None if (device is None or _host_backend()) else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        pend = int(t.item())
    if pend > 0:
        # the profiles of those rows, over all ranks; every rank counts them exactly; the counters meet again
        f = torch.from_numpy(engine.lazy_pending_profiles().astype(np.int32))
        if dist.is_available() and dist.is_initialized():
            if device is not None and not _host_backend():
                f = f.to(device)
            dist.all_reduce(f, op=dist.ReduceOp.MAX)
        engine.lazy_complete(f.cpu().numpy())
        allreduce_domz_device(engine, device)
        engine.finalize(domE=domE)
        left = engine.lazy_pending()
        if dist.is_available() and dist.is_initialized():
            t = torch.tensor([left], dtype=torch.int64, device=None if (device is None or _host_backend()) else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            left = int(t.item())
        if left > 0:                       # the safety net (a counted profile leaves no row undecided): everything in full, on every rank
            if search is None:
                raise RuntimeError("%d domain rows of the lazy search stayed undecided and no `search` callable was given to repeat it" % left)
            keep = engine.rows_mode
            engine.set_rows_mode("compact")
            try:
                search()
                allreduce_domz_device(engine, device)
                engine.finalize(domE=domE)
            finally:
                engine.set_rows_mode(keep)
    return pend


# ------------------------------------------------------------------------------------------------
# Exact dereplication across shards (option 2): hash-partitioned all-to-all.
#
# Each rank dereplicates its own shard on its GPU (exact, verified word by word).  Then only the UNIQUES meet: per unique a
# 128-bit key (two XXH64 seeds over the packed read incl. its length), made orientation-free by taking the smaller of
# (forward, reverse complement) -- vsearch --strand both joins a read to a seed that equals it or its reverse complement.
# Every key has an OWNER rank (key mod world): one all-to-all carries (key, global index of the first occurrence, orientation
# flag, local unique number) to the owners -- 40 B per unique, each rank receives ~1/world of them, nothing is replicated.
# The owner groups its keys (one sort of its share) and decides, per distinct sequence: the global first occurrence (the
# representative, whose orientation the cluster keeps) and the SCORER -- one of the ranks that hold the sequence in the
# representative's orientation, picked by the key, so that the scoring work is spread evenly even when the shards share most
# of their sequences (with "the first occurrence scores" the low ranks did all of it).  A second all-to-all returns the
# verdicts.  After the search the coordinates travel the same way: holders ask the scorer (one all-to-all of unique numbers,
# one of int32 x 4 rows).  Result = what one GPU computes on the concatenated input: same representatives (and orientation),
# same domZ, same per-read coordinates.  Cross-rank equality is by 128-bit key, not re-verified.
# All of it runs on tensors that live where the engine left them (device memory under RCCL, host memory under gloo).
def _a2a_var(rows, dest, world):
    """rows [n, w] to the ranks in dest [n] -> (received rows, recv counts, send counts, the permutation that grouped them)"""
    import torch
    import torch.distributed as dist
    order = torch.argsort(dest, stable=True)
    send = rows[order].contiguous()
    sc = torch.bincount(dest, minlength=world).to(torch.int64)
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc)
    rcl, scl = rc.cpu().tolist(), sc.cpu().tolist()
    out = torch.empty((int(sum(rcl)),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    dist.all_to_all_single(out, send, output_split_sizes=rcl, input_split_sizes=scl)
    return out, rcl, scl, order


def _a2a_back(rows, rcl, scl, order):
    """answers to the rows received by _a2a_var, returned to their senders in the senders' original order"""
    import torch
    import torch.distributed as dist
    back = torch.empty((int(sum(scl)),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    dist.all_to_all_single(back, rows.contiguous(), output_split_sizes=scl, input_split_sizes=rcl)
    out = torch.empty_like(back)
    out[order] = back
    return out


def owner_verdicts(recv, src):
    """The owner's step (pure; also run by the CPU tests): recv [m, 5] int64 = (key0, key1, gidx, fwd flag, local unique number)
    from ranks src [m].  Returns [m, 4] int64 per row: gidx and orientation flag of the group's global first occurrence,
    rank and local unique number of the holder that scores the sequence."""
    import torch
    m = recv.shape[0]
    if m == 0:
        return torch.zeros((0, 4), dtype=torch.int64, device=recv.device)
    order = torch.argsort(recv[:, 2], stable=True)
    order = order[torch.argsort(recv[order, 1], stable=True)]
    order = order[torch.argsort(recv[order, 0], stable=True)]
    s, ssrc = recv[order], src[order]
    first = torch.ones(m, dtype=torch.bool, device=recv.device)
    first[1:] = (s[1:, 0] != s[:-1, 0]) | (s[1:, 1] != s[:-1, 1])
    grp = torch.cumsum(first.to(torch.int64), 0) - 1
    head = torch.nonzero(first).flatten()
    seed_gidx, seed_fwd = s[head, 2][grp], s[head, 3][grp]
    cand = (s[:, 3] == seed_fwd).to(torch.int64)                      # holders of the sequence in the representative's orientation
    run = torch.cumsum(cand, 0) - cand                                # candidates before this row
    pos = run - run[head][grp]                                        # ... inside the group
    cnt = torch.zeros(head.shape[0], dtype=torch.int64, device=recv.device).index_add_(0, grp, cand)
    pick = torch.remainder(s[head, 1], cnt)[grp]                      # by the key: even over the holders
    chosen = (cand == 1) & (pos == pick)
    sr = torch.zeros(head.shape[0], dtype=torch.int64, device=recv.device)
    su = torch.zeros(head.shape[0], dtype=torch.int64, device=recv.device)
    sr[grp[chosen]] = ssrc[chosen]
    su[grp[chosen]] = s[chosen, 4]
    ans = torch.stack([seed_gidx, seed_fwd, sr[grp], su[grp]], dim=1)
    out = torch.empty_like(ans)
    out[order] = ans
    return out


def global_derep(engine, n_reads_local, device=None):
    """After engine.derep(): match the local uniques against every other rank's (hash-partitioned, see above).  Marks the
    uniques another rank scores inactive (engine.set_active_uniques) and returns the bookkeeping exchange_coords() needs.
    Without an initialised process group this is a no-op."""
    import torch
    import torch.distributed as dist
    import os
    U = engine.n_unique
    force = os.environ.get("ITSX_FORCE_DIST") == "1"          # one rank, every collective all the same (the nccl test on a one-GPU box)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        seed_read, _ = engine.get_uniques()
        return dict(active=np.ones(U, bool), seed_gidx=seed_read.astype(np.int64), flip=np.zeros(U, bool), base=0, scorer=None)
    ws, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([int(n_reads_local)], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(ns, n)
    base = int(sum(int(x.item()) for x in ns[:rank]))
    tup = engine.unique_tuples(base, device)                          # [U, 4]: key0, key1, gidx, fwd flag -- where the engine left them
    if tup.is_cuda and _host_backend():
        tup = tup.cpu()
    lu = torch.arange(U, dtype=torch.int64, device=tup.device)
    rows = torch.cat([tup, lu[:, None]], dim=1)
    dest = torch.remainder(tup[:, 0], ws)
    recv, rcl, scl, order = _a2a_var(rows, dest, ws)
    src = torch.repeat_interleave(torch.arange(ws, dtype=torch.int64, device=tup.device), torch.tensor(rcl, dtype=torch.int64, device=tup.device))
    verdict = _a2a_back(owner_verdicts(recv, src), rcl, scl, order)   # [U, 4]: seed gidx, seed fwd, scorer rank, scorer's unique number
    active = (verdict[:, 2] == rank) & (verdict[:, 3] == lu)
    engine.set_active_uniques(active.cpu().numpy())
    return dict(active=active.cpu().numpy(), seed_gidx=verdict[:, 0].cpu().numpy(), flip=(verdict[:, 1] != tup[:, 3]).cpu().numpy(), base=base,
                scorer=verdict[:, 2:4], active_t=active)


def exchange_rows(g, rep_rows):
    """After finalize(): rep_rows = [U, 4] int32 rows per local unique (engine.rep_coords_device).  The uniques another rank
    scored get that rank's rows: one all-to-all of requests (the scorer's unique numbers), one of answers."""
    import torch
    import torch.distributed as dist
    import os
    if g.get("scorer") is None or not (dist.is_available() and dist.is_initialized()) or \
            (dist.get_world_size() == 1 and os.environ.get("ITSX_FORCE_DIST") != "1"):
        return rep_rows
    ws = dist.get_world_size()
    if rep_rows.device != g["active_t"].device:
        rep_rows = rep_rows.to(g["active_t"].device)
    need = ~g["active_t"]
    req = g["scorer"][need]
    recv, rcl, scl, order = _a2a_var(req[:, 1:2].contiguous(), req[:, 0].contiguous(), ws)
    ans = rep_rows[recv[:, 0]]
    back = _a2a_back(ans, rcl, scl, order)
    out = rep_rows.clone()
    out[need] = back
    return out


def exchange_coords(g, start, stop, tlen, ind, device=None):
    """numpy front end of exchange_rows (tests, file-compatible callers): four arrays per local unique in, four out."""
    import torch
    rows = torch.from_numpy(np.stack([start, stop, tlen, ind], axis=1).astype(np.int32))
    if device is not None:
        rows = rows.to(device)
    out = exchange_rows(g, rows).cpu().numpy()
    return tuple(out[:, k].astype(a.dtype) for k, a in enumerate((start, stop, tlen, ind)))


def read_rows(engine, rep_rows, device=None):
    """per-read rows from per-representative rows, on the device: rows[uniq_of], (-1, -1, -1, 0) for dropped reads"""
    import torch
    uq = engine.derep_device(device)["uniq_of"].to(torch.int64)
    if rep_rows.device != uq.device:          # after a host-side exchange (gloo): back to where the reads' map lives
        rep_rows = rep_rows.to(uq.device)
    ok = uq >= 0
    none = torch.tensor([-1, -1, -1, 0], dtype=torch.int32, device=rep_rows.device)
    if rep_rows.shape[0] == 0:                # a shard whose reads were all dropped (no unique at all): what itsx_trim_coords returns
        return none[None, :].expand(uq.shape[0], 4).contiguous()
    rows = rep_rows[uq.clamp(min=0)]
    return torch.where(ok[:, None], rows, none[None, :])
