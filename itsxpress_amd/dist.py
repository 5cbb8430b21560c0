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


# ------------------------------------------------------------------------------------------------
# Exact dereplication across shards (option 2).
#
# Each rank dereplicates its own shard on its GPU (exact, verified word by word), then only the UNIQUES are
# matched across ranks: per unique a 128-bit key (two XXH64 seeds over the packed read incl. its length), made
# orientation-free by taking the smaller of (forward, reverse-complement) -- vsearch --strand both joins a read
# to a seed that equals it or its reverse complement.  Every rank gets all keys (all_gather: 24 B per unique),
# groups them with one sort, and learns for each of its uniques which rank holds the GLOBAL first occurrence.
# That rank scores the sequence (it has the bases); the others mark the unique inactive and receive its
# coordinates afterwards.  Result = what one GPU computes on the concatenated input: same representatives (and
# orientation), same domZ, same per-read coordinates.  Cross-rank equality is by 128-bit key, not re-verified.
_SEED_A, _SEED_B = 0x1F83D9ABFB41BD6B, 0x5BE0CD19137E2179


def _all_gather_var(t, device):
    """all_gather of 1-D/2-D tensors whose first dimension differs between ranks -> list of tensors."""
    import torch
    import torch.distributed as dist
    ws = dist.get_world_size()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(max(sizes), 1)
    pad = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=device)
    pad[:t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(out, pad)
    return [o[:s] for o, s in zip(out, sizes)]


def group_keys(c0, c1, gidx, orient):
    """Pure grouping step (also used by the CPU tests): int64 tensors of equal length.
    Returns (seed_gidx, seed_orient) per element: the smallest gidx with the same (c0, c1) and its orient."""
    import torch
    n = c0.shape[0]
    if n == 0:
        return gidx.clone(), orient.clone()
    order = torch.argsort(gidx, stable=True)
    order = order[torch.argsort(c1[order], stable=True)]
    order = order[torch.argsort(c0[order], stable=True)]
    s0, s1 = c0[order], c1[order]
    first = torch.ones(n, dtype=torch.bool, device=c0.device)
    first[1:] = (s0[1:] != s0[:-1]) | (s1[1:] != s1[:-1])
    grp = torch.cumsum(first.to(torch.int64), 0) - 1
    head = torch.nonzero(first).flatten()                     # position (in sorted order) of each group's first element
    seed_sorted = gidx[order][head][grp]
    orient_sorted = orient[order][head][grp]
    seed = torch.empty_like(gidx)
    so = torch.empty_like(orient)
    seed[order] = seed_sorted
    so[order] = orient_sorted
    return seed, so


def global_derep(engine, n_reads_local, device=None):
    """After engine.derep(): match the local uniques against every other rank's.  Marks the uniques whose global
    first occurrence lives elsewhere inactive (engine.set_active_uniques) and returns the bookkeeping
    exchange_coords() needs.  Without an initialised process group this is a no-op."""
    import torch
    import torch.distributed as dist
    U = engine.n_unique
    seed_read, _ = engine.get_uniques()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(active=np.ones(U, bool), gidx=seed_read.astype(np.int64), seed_gidx=seed_read.astype(np.int64),
                    flip=np.zeros(U, bool), base=0)
    rank = dist.get_rank()
    n = torch.tensor([int(n_reads_local)], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n) for _ in range(dist.get_world_size())]
    dist.all_gather(ns, n)
    base = int(sum(int(x.item()) for x in ns[:rank]))
    kf0, kr0 = engine.unique_keys(_SEED_A)
    kf1, kr1 = engine.unique_keys(_SEED_B)
    fwd_le = (kf0 < kr0) | ((kf0 == kr0) & (kf1 <= kr1))     # is the read in its canonical orientation?
    c0 = np.where(fwd_le, kf0, kr0).view(np.int64)
    c1 = np.where(fwd_le, kf1, kr1).view(np.int64)
    gidx = seed_read.astype(np.int64) + base
    loc = torch.from_numpy(np.stack([c0, c1, gidx, fwd_le.astype(np.int64)], axis=1).copy())
    if device is not None:
        loc = loc.to(device)
    parts = _all_gather_var(loc, device)
    allk = torch.cat(parts, 0)
    seed, so = group_keys(allk[:, 0], allk[:, 1], allk[:, 2], allk[:, 3])
    lo = sum(p.shape[0] for p in parts[:rank])
    seed = seed[lo:lo + U].cpu().numpy()
    so = so[lo:lo + U].cpu().numpy()
    active = seed == gidx
    engine.set_active_uniques(active)
    return dict(active=active, gidx=gidx, seed_gidx=seed, flip=so.astype(bool) != fwd_le, base=base)


def exchange_coords(g, start, stop, tlen, ind, device=None):
    """After finalize(): start/stop/tlen/ind per local unique (engine.rep_coords).  Fills the inactive uniques from
    the rank that scored their global first occurrence; returns the four completed arrays."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return start, stop, tlen, ind
    act = g["active"]
    mine = np.stack([g["gidx"][act], start[act].astype(np.int64), stop[act].astype(np.int64), tlen[act].astype(np.int64),
                     ind[act].astype(np.int64)], axis=1) if act.any() else np.zeros((0, 5), np.int64)
    t = torch.from_numpy(np.ascontiguousarray(mine))
    if device is not None:
        t = t.to(device)
    tab = torch.cat(_all_gather_var(t, device), 0).cpu().numpy()
    tab = tab[np.argsort(tab[:, 0], kind="stable")]
    need = ~act
    out = [start.copy(), stop.copy(), tlen.copy(), ind.copy()]
    if need.any():
        pos = np.searchsorted(tab[:, 0], g["seed_gidx"][need])
        assert np.array_equal(tab[pos, 0], g["seed_gidx"][need]), "a global representative was scored by no rank"
        for k in range(4):
            out[k][need] = tab[pos, 1 + k].astype(out[k].dtype)
    return tuple(out)
