"""N>1 path on CPU: world_size-2 gloo processes exercise the two exchange steps of the path
(all-reduce of domZ, gather of per-read coordinates) and the shard arithmetic."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from itsxpress_amd.dist import allreduce_domz, gather_coords, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total = 1001
    lo, hi = shard_bounds(n_total, world, rank)
    z = allreduce_domz(np.arange(7, dtype=np.int64) * (rank + 1))
    idx = np.arange(lo, hi, dtype=np.int32)
    out = gather_coords(idx, idx + 1, idx + 2, (idx % 2).astype(np.int32))
    if rank == 0:
        allc = np.concatenate(out)
        q.put((z.tolist(), allc.shape, bool(np.array_equal(allc[:, 0], np.arange(n_total))),
               bool(np.array_equal(allc[:, 2], np.arange(n_total) + 2))))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_steps():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    z, shape, ok0, ok2 = res
    assert z == [0, 3, 6, 9, 12, 15, 18]
    assert tuple(shape) == (1001, 4) and ok0 and ok2


def test_shard_bounds_cover_everything_in_order():
    from itsxpress_amd.dist import shard_bounds
    for n in (0, 1, 7, 1000003):
        for w in (1, 2, 4, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
