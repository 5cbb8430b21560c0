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
    """stands in for the GPU engine: the uniques of a shard, "sequences" being integers s whose reverse complement is the
    sequence 1000 - s; the orientation-free key is the smaller of the two"""

    def __init__(self, seqs, seed_read):
        self.n_unique = len(seqs)
        self._seqs = np.asarray(seqs, np.int64)
        self._seed_read = np.asarray(seed_read, np.int64)
        self.active = None

    def get_uniques(self):
        return self._seed_read, np.ones(self.n_unique, np.int64)

    def unique_tuples(self, base, device=None):
        import torch
        s = self._seqs
        canon = np.minimum(s, 1000 - s)
        rows = np.stack([canon * 2654435761 % 1000003 - 500000, canon * 7 + 1, self._seed_read + base, (s <= 1000 - s).astype(np.int64)], axis=1)
        return torch.from_numpy(rows.astype(np.int64))

    def set_active_uniques(self, active):
        self.active = np.asarray(active, bool)


def _shards(world):
    """per rank: (sequences of its uniques, local read index of each one's first occurrence, reads in the shard)"""
    if world == 2:       # shard 0 holds 5, 7, 990 ; shard 1 holds 7, 995 (= rc of 5), 10 (= rc of 990), 3
        return [([5, 7, 990], [0, 2, 3], 6), ([7, 995, 10, 3], [0, 1, 4, 5], 8)]
    rng = np.random.default_rng(11)
    pool = rng.choice(np.arange(1, 500), 120, replace=False)           # 120 distinct sequences, none its own reverse complement
    out = []
    for r in range(world):
        pick = rng.choice(pool, 70, replace=False)
        seqs = np.where(rng.random(70) < 0.3, 1000 - pick, pick)        # some of them held reverse-complemented
        out.append((seqs.tolist(), sorted(rng.choice(400, 70, replace=False).tolist()), 400))
    return out


def _worker_global(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from itsxpress_amd.dist import exchange_rows, global_derep
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs, seed_read, n_local = _shards(world)[rank]
    eng = _FakeEngine(seqs, seed_read)
    g = global_derep(eng, n_local)
    # every rank's rows say who computed them: (rank, unique number, sequence, 1); only scored uniques have rows
    rows = torch.tensor([[rank, u, s, 1] if g["active"][u] else [-9, -9, -9, 0] for u, s in enumerate(seqs)], dtype=torch.int32)
    out = exchange_rows(g, rows)
    q.put((rank, eng.active.tolist(), g["seed_gidx"].tolist(), g["flip"].tolist(), out.tolist(), g["base"]))
    dist.barrier()
    dist.destroy_process_group()


def _run_global(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_global, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(world):
        r = q.get(timeout=180)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _check_global(world, res):
    shards = _shards(world)
    bases = np.cumsum([0] + [s[2] for s in shards])
    # what one process would decide: per canonical sequence the global first occurrence and its orientation
    first = {}
    for r, (seqs, seed_read, _) in enumerate(shards):
        for u, s in enumerate(seqs):
            c = min(s, 1000 - s)
            gi = int(bases[r] + seed_read[u])
            if c not in first or gi < first[c][0]:
                first[c] = (gi, s)
    scorers = {}
    for r, (seqs, seed_read, _) in enumerate(shards):
        active, seed_gidx, flip, rows, base = res[r]
        assert base == bases[r]
        for u, s in enumerate(seqs):
            c = min(s, 1000 - s)
            assert seed_gidx[u] == first[c][0]                      # the representative is the global first occurrence
            assert flip[u] == (s != first[c][1])                    # ... and a holder of the other orientation is flipped
            if active[u]:
                assert s == first[c][1]                             # the scorer holds the representative's orientation
                scorers.setdefault(c, []).append((r, u))
    assert all(len(v) == 1 for v in scorers.values()) and set(scorers) == set(first)   # exactly one scorer per sequence
    for r, (seqs, _, _) in enumerate(shards):
        rows = res[r][3]
        for u, s in enumerate(seqs):
            sr, su = scorers[min(s, 1000 - s)][0]
            assert rows[u] == [sr, su, first[min(s, 1000 - s)][1], 1]            # everybody ends up with the scorer's rows
    return scorers


def test_two_rank_global_derep():
    res = _run_global(2)
    _check_global(2, res)
    # shard 1 (global index base 6): 7 -> seed at global 2 (same strand); 995 -> seed 5 at global 0, reverse complement;
    # 10 -> seed 990 at global 3, reverse complement; 3 is new
    assert res[1][1] == [2, 0, 3, 11] and res[1][2] == [False, True, True, False]


def test_four_rank_global_derep_spreads_the_scoring():
    """hash-partitioned matching on four ranks: one scorer per distinct sequence, in the representative's orientation, and
    the scoring work is spread over the holders instead of piling onto the ranks with the first occurrences"""
    res = _run_global(4)
    scorers = _check_global(4, res)
    load = np.bincount([v[0][0] for v in scorers.values()], minlength=4)
    assert load.min() >= 0.5 * load.mean(), load


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

    def run(*flags):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-selftest"] + list(flags), env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])

    # the driver passes only --gpus N: that must be the honest curve, configs[2]'s 10 M reads sharded over the ranks (strong scaling)
    rec = run()
    assert rec["n_gpus"] == 2 and rec["ok"] is True
    assert rec["scaling"] == "strong" and rec["total_reads"] == 10000000 and rec["reads_rank0"] == 5000000 and rec["reads_all_ranks"] == 10000000
    # weak scaling is opt-in
    rec = run("--weak")
    assert rec["scaling"] == "weak" and rec["reads_rank0"] == 10000000 and rec["reads_all_ranks"] == 20000000
    rec = run("--total-reads", "3000001")
    assert rec["scaling"] == "strong" and rec["reads_all_ranks"] == 3000001
