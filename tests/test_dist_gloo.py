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


# ---------------------------------------------------------------- exact dereplication across shards (SURVEY 8e option 2)
class _FakeEngine:
    """stands in for the GPU engine: uniques of a shard with made-up 64-bit keys (forward / reverse complement)"""

    def __init__(self, seed_read, kf, kr):
        self.n_unique = len(seed_read)
        self._seed_read = np.asarray(seed_read, np.int64)
        self._kf = np.asarray(kf, np.uint64)
        self._kr = np.asarray(kr, np.uint64)
        self.active = None

    def get_uniques(self):
        return self._seed_read, np.ones(self.n_unique, np.int64)

    def unique_keys(self, seed):
        m = np.uint64(seed & 0xFFFF)
        return self._kf * np.uint64(3) + m, self._kr * np.uint64(3) + m

    def set_active_uniques(self, active):
        self.active = np.asarray(active, bool)


def _worker_global(rank, world, port, q):
    import torch.distributed as dist
    from itsxpress_amd.dist import exchange_coords, global_derep
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # sequences are integers; key of sequence s = (s, 1000 - s) for (forward, reverse complement); the reverse
    # complement of s is the sequence 1000 - s.  shard 0 holds 5, 7, 990 ; shard 1 holds 7, 995 (= rc of 5), 10, 3
    if rank == 0:
        seqs, seed_read, n_local = [5, 7, 990], [0, 2, 3], 6
    else:
        seqs, seed_read, n_local = [7, 995, 10, 3], [0, 1, 4, 5], 8
    eng = _FakeEngine(seed_read, seqs, [1000 - s for s in seqs])
    g = global_derep(eng, n_local)
    start = np.array([100 + s for s in seqs], np.int32)
    start[~g["active"]] = -1
    out = exchange_coords(g, start, start + 1, start + 2, (start >= 0).astype(np.int32))
    q.put((rank, eng.active.tolist(), g["seed_gidx"].tolist(), g["flip"].tolist(), out[0].tolist(), out[1].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_global_derep():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_global, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(world):
        r = q.get(timeout=120)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # shard 0: every unique is a global first occurrence
    assert res[0][0] == [True, True, True] and res[0][1] == [0, 2, 3] and res[0][2] == [False, False, False]
    # shard 1 (global index base 6): 7 -> seed at global 2 (same strand); 995 -> seed 5 at global 0, reverse complement;
    # 10 -> seed 990 at global 3, reverse complement; 3 is new
    assert res[1][0] == [False, False, False, True]
    assert res[1][1] == [2, 0, 3, 11] and res[1][2] == [False, True, True, False]
    # coordinates of the inactive uniques come from the rank that scored the seed
    assert res[1][3] == [107, 105, 1090, 103] and res[1][4] == [108, 106, 1091, 104]


def test_assign_samples_balances_whole_samples():
    from itsxpress_amd.dist import assign_samples
    rng = np.random.default_rng(3)
    sizes = rng.integers(100, 200000, 96)
    for ws in (1, 2, 8):
        parts = assign_samples(sizes, ws)
        assert sorted(i for p in parts for i in p) == list(range(96))          # every sample exactly once
        loads = [int(sizes[p].sum()) for p in parts]
        assert max(loads) - min(loads) <= int(sizes.max())                       # LPT: within one sample of even
    assert assign_samples([5, 1], 4) == [[0], [1], [], []]


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (no launcher, no WORLD_SIZE) must itself start two ranks and print one line with
    n_gpus = 2: checked here with the engine-free self-test leg over gloo (the real leg needs GPUs)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-selftest"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ok"] is True
