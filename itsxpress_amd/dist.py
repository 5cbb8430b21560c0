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


def gather_coords(start, stop, tlen, in_ddict, device=None, dst=0):
    """rank dst receives every rank's [n_i, 4] int32 block (shards may differ in size)."""
    import torch
    import torch.distributed as dist
    block = np.stack([start, stop, tlen, in_ddict], axis=1).astype(np.int32)
    if not (dist.is_available() and dist.is_initialized()):
        return [block]
    ws, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([block.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    nmax = int(max(int(s.item()) for s in sizes))
    pad = np.full((nmax, 4), -1, np.int32)
    pad[:block.shape[0]] = block
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    # all_gather (the collective every backend implements) rather than gather: the blocks are tiny next to the
    # DP work (16 B per read), and rank dst simply keeps what it needs
    out = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(out, t)
    if rank != dst:
        return None
    return [o.cpu().numpy()[:int(s.item())] for o, s in zip(out, sizes)]
