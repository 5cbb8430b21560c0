"""Two ranks (gloo), one GPU: the sharded path with exact cross-shard dereplication (SURVEY 8e option 2) must give,
read for read, what ONE engine computes on the concatenated input -- representatives, strands, coordinates.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import synth

pytestmark = pytest.mark.gpu
_RC = str.maketrans("ACGTN", "TGCAN")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _library(hmm_text):
    blob, offs = synth.make_reads(hmm_text, 3000, seed=77, sub_rate=0.002)
    reads = synth.to_strings(blob, offs)
    # make sure the shards share sequences in both orientations: copy some of the first half into the second, some reversed
    n = len(reads)
    for k in range(0, 600):
        src = reads[(k * 7) % (n // 2)]
        reads[n // 2 + k] = src if k % 3 else src[::-1].translate(_RC)
    return reads


def _run_path(eng, reads, hmm_text, g_fn=None, x_fn=None, domz_fn=None):
    eng.load_profiles(text=hmm_text)
    eng.set_reads(reads)
    eng.derep()
    g = g_fn(eng, len(reads)) if g_fn else None
    eng.search()
    if domz_fn:
        eng.set_domz(domz_fn(eng.get_domz()))
    eng.finalize()
    us, ue, ut, ui = eng.rep_coords("3_", "4_")
    if x_fn:
        us, ue, ut, ui = x_fn(g, us, ue, ut, ui)
    rep_of, strand, uniq_of = eng.get_derep()
    ok = uniq_of >= 0
    start = np.where(ok, us[np.maximum(uniq_of, 0)], -1)
    stop = np.where(ok, ue[np.maximum(uniq_of, 0)], -1)
    tlen = np.where(ok, ut[np.maximum(uniq_of, 0)], -1)
    return g, start, stop, tlen, rep_of, strand, uniq_of


def _run_path_device(eng, reads, hmm_text):
    """the same path through the device-resident entry points bench.py uses at N > 1: the counters are reduced and the
    coordinates exchanged / fanned out to the reads in the engine's own device buffers"""
    from itsxpress_amd.dist import allreduce_domz_device, exchange_rows, global_derep, read_rows
    eng.load_profiles(text=hmm_text)
    eng.set_reads(reads)
    eng.derep()
    g = global_derep(eng, len(reads))
    eng.search()
    allreduce_domz_device(eng)
    eng.finalize()
    rows = read_rows(eng, exchange_rows(g, eng.rep_coords_device("3_", "4_")).to("cuda")).cpu().numpy()
    rep_of, strand, uniq_of = eng.get_derep()
    return g, rows[:, 0], rows[:, 1], rows[:, 2], rep_of, strand, uniq_of


def _worker(rank, world, port, hmm_text, reads, q):
    import torch.distributed as dist
    from itsxpress_amd import Engine
    from itsxpress_amd.dist import allreduce_domz, exchange_coords, global_derep, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(len(reads), world, rank)
    eng = Engine(0)
    g, start, stop, tlen, rep_of, strand, uniq_of = _run_path(eng, reads[lo:hi], hmm_text, global_derep, exchange_coords, allreduce_domz)
    g2, start2, stop2, tlen2, _, _, _ = _run_path_device(eng, reads[lo:hi], hmm_text)
    assert np.array_equal(start, start2) and np.array_equal(stop, stop2) and np.array_equal(tlen, tlen2)
    assert np.array_equal(g["seed_gidx"], g2["seed_gidx"]) and np.array_equal(g["active"], g2["active"])
    ok = uniq_of >= 0
    grep = np.where(ok, g["seed_gidx"][np.maximum(uniq_of, 0)], -1)
    gstrand = np.where(ok & g["flip"][np.maximum(uniq_of, 0)], -strand, strand)
    q.put((rank, start, stop, tlen, grep, gstrand, int(g["active"].sum()), int(eng.n_unique)))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_two_ranks_equal_one_engine(engine, mini_hmm_text):
    reads = _library(mini_hmm_text)
    _, start, stop, tlen, rep_of, strand, _ = _run_path(engine, reads, mini_hmm_text)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mini_hmm_text, reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=600)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cat = lambda k: np.concatenate([res[0][k], res[1][k]])
    assert np.array_equal(cat(0), start) and np.array_equal(cat(1), stop) and np.array_equal(cat(2), tlen)
    assert np.array_equal(cat(3), rep_of) and np.array_equal(cat(4), strand)
    # both shards left sequences to the other one (the scorer of a shared sequence is picked by its key, not by shard order)
    assert res[1][5] < res[1][6] and res[0][5] < res[0][6] and (start >= 0).sum() > 2000


def test_bench_runs_its_n_rank_path_with_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` starts two ranks itself; here both share GPU 0 and talk over gloo (ITSX_BENCH_ONE_GPU /
    ITSX_BENCH_BACKEND: RCCL needs a GPU per rank), so the step's N > 1 branch -- device-resident domZ all-reduce, gather of
    the coordinate rows, and with --global-derep the hash-partitioned matching -- runs end to end and prints ONE line"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(ITSX_BENCH_ONE_GPU="1", ITSX_BENCH_BACKEND="gloo")
    for extra in (["--reads", "40000", "--per-shard-derep"], ["--total-reads", "80000"]):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "cfg1", "--steps", "1", "--warmup", "1",
                            "--cpu-sample", "0", "--handover-steps", "0"] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["config"]["reads_trimmed_rank0"] > 30000
        assert rec["scaling"] == ("strong" if "--total-reads" in extra else "weak")
        conc = rec["concordance_vs_single_engine"]
        if "--per-shard-derep" not in extra:          # the default: exact global dereplication == one engine on the whole job, read for read
            assert conc["equal"] is True and conc["reads"] == 80000 and "exact global" in conc["derep"]
        else:
            assert conc["fraction_of_reads_equal"] > 0.99


def test_every_collective_over_rccl_with_one_rank():
    """the `nccl` backend (= RCCL) on this one-GPU box: one rank, ITSX_FORCE_DIST=1 -- init_process_group(device_id=), the in-place
    all-reduce on the engine's zero-copy counter view, the all-to-all of the unique keys and of the coordinate rows (int64 / int32),
    the gather of the rows -- and the line's own check that the result is the one-engine result"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "ITSX_BENCH_BACKEND", "ITSX_BENCH_ONE_GPU")}
    env.update(ITSX_FORCE_DIST="1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "cfg1", "--reads", "60000", "--steps", "1", "--warmup", "1",
                        "--cpu-sample", "0", "--handover-steps", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    rec = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["ranks"] is not None and rec["ranks"]["allreduce_ms_per_step_max"] > 0
    assert rec["config"]["parallelism"].endswith("exact global derep")
    assert rec["concordance_vs_single_engine"]["equal"] is True
    assert rec["full_pipeline"]["coordinates_equal_lazy"] is True
