#!/usr/bin/env python3
"""bench.py -- reads/sec trimmed on the hot path (BASELINE.json metric), 1..8 GPUs of one node.

A "step" is one pass of the hot path over one batch of synthetic reads already packed and
resident in HBM: derep -> MSV -> bias/Forward -> Backward/domain definition -> envelope
re-scoring -> [all-reduce domZ] -> thresholds/argmax -> per-read (start, stop, tlen) on the
host of every rank -> gather to rank 0.  Workload at N=1: BASELINE.json configs[1]
(1M synthetic 300 bp single-end reads, ITS2, cluster_id=1.0); Fungi's model file is absent
from the reference mount, so the stand-in taxon Tracheophyta (155 ITS2 profiles) is used and
labelled.  Weak scaling: every rank processes its own shard of that size.

Prints ONE JSON line on rank 0.
"""
import argparse
import gzip
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak (MI355X_MICROARCH.md)
# HBM bytes per lane-row measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), see
# profiles/round1_pmc_hbm_traffic_200k.md.  WRITE_SIZE is exact for these stores; FETCH_SIZE is NOT doubled
# (uncalibrated for 4-byte-per-lane loads on gfx950), so the read side is a lower bound.
PMC_BYTES_PER_ROW = {"k_filters_fwd": 0.6 + 23.7, "k_bwd_decode": 13.2 + 23.6, "k_decode": 17.8 + 0.2}
VALU_PEAK_GOPS = 256 * 4 * 32 * 2.4   # lane-ops/ns: 256 CUs x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s


def its2_profiles(hmm_text):
    blocks = [b + "//\n" for b in hmm_text.split("//\n") if "NAME  " in b]
    return "".join(b for b in blocks if b.split("NAME  ")[1][:2] in ("3_", "4_"))


def cpu_baseline(hmm_its2, blob, offs, sample_reads, threads):
    """the oracle (a port of the reference's CPU path) timed on a bounded sample of the same workload"""
    import orc
    seqs = [blob[offs[i]:offs[i + 1]].decode() for i in range(sample_reads)]
    hs = orc.HmmSet(text=hmm_its2)
    t0 = time.time()
    codes, o = orc.digitize(seqs)
    nc, rep, strand = orc.derep(codes, o)
    seeds = [i for i in range(len(seqs)) if rep[i] == i]
    c2, o2 = orc.digitize([seqs[i] for i in seeds])
    res = orc.SearchResult(hs, c2, o2, threads=threads, keep_trace=0)
    us, ue, ut, ui = res.positions("3_", "4_")
    dt = time.time() - t0
    uniq = np.cumsum(np.asarray(rep) == np.arange(len(seqs))) - 1          # unique index of each seed, in input order
    uo = uniq[np.asarray(rep)]
    coords = np.stack([us[uo], ue[uo], ut[uo]], axis=1)                      # per read of the sample: the baseline path's answer
    return sample_reads / dt, dt, nc, coords


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1000000, help="reads per GPU")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads in the CPU-baseline sample (0 = skip, -1 = auto: ~20 s of CPU work)")
    ap.add_argument("--cluster-id", type=float, default=1.0,
                    help="1.0 = exact dereplication (BASELINE configs[1], the default); < 1 runs row a2 (greedy clustering) instead")
    ap.add_argument("--taxa", choices=["T", "all"], default="T",
                    help="T = the stand-in taxon of BASELINE configs[1] (155 ITS2 profiles); all = --taxa All --region ITS2 (814 profiles, configs[3])")
    ap.add_argument("--global-derep", action="store_true",
                    help="N > 1: match the uniques across shards (exact global dereplication, SURVEY 8e option 2) instead of per-shard")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("ITSX_FORCE_DIST") == "1"     # the latter: exercise RCCL with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from itsxpress_amd import Engine
    from itsxpress_amd.dist import allreduce_domz, exchange_coords, gather_coords, global_derep
    import synth

    with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f:
        thmm = f.read()
    hmm = its2_profiles(thmm)
    if args.taxa == "all":
        with gzip.open(os.path.join(ROOT, "tests", "golden", "all_its2.hmm.gz"), "rt") as f:
            hmm = f.read()
    blob, offs = synth.make_reads(thmm, args.reads, config=2, seed=synth.SEED + 2 + 1000 * rank)
    eng = Engine(local_rank)
    nprof = eng.load_profiles(text=hmm)
    eng.set_reads_buffer(blob, offs)          # pack + upload: inputs are resident in HBM before timing

    def step():
        if args.cluster_id < 1.0:
            eng.cluster(args.cluster_id, strand_both=True)
        else:
            eng.derep(strand_both=True, minseqlength=32)
        g = global_derep(eng, args.reads, dev) if (use_dist and args.global_derep) else None
        eng.search(T=10.0, F1=1e-6, F2=1e-6, F3=1e-6)
        if use_dist:
            eng.set_domz(allreduce_domz(eng.get_domz(), dev))
        eng.finalize(domE=10.0)
        if g is not None:            # coordinates of the uniques scored elsewhere arrive here, then fan out to the reads
            us, ue, ut, ui = exchange_coords(g, *eng.rep_coords("3_", "4_"), device=dev)
            uq = eng.get_derep()[2]
            ok = uq >= 0
            uq = np.maximum(uq, 0)
            c = tuple(np.where(ok, a[uq], d).astype(np.int32) for a, d in ((us, -1), (ue, -1), (ut, -1), (ui, 0)))
        else:
            c = eng.trim_coords("3_", "4_")
        return gather_coords(*c, device=dev) if use_dist else [np.stack(c, axis=1)]

    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = {}
    out = None
    for _ in range(args.steps):
        out = step()
        st = eng.stats()
        for k, v in st.items():
            if k.startswith("ms_"):
                acc[k] = acc.get(k, 0.0) + v
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = eng.stats()

    if rank == 0:
        total_reads = args.reads * world * args.steps
        value = total_reads / dt
        K = args.steps
        kern = {"k_msv": acc["ms_msv_kernel"] / K, "k_filters_fwd": acc["ms_fwd_kernel"] / K,
                "k_bwd_decode": acc["ms_bwd_kernel"] / K, "k_decode": acc["ms_decode_kernel"] / K,
                "k_env_fwd+k_env_bwd+k_env_post": acc["ms_env_kernel"] / K}
        dom = max(kern, key=kern.get)
        # algorithmic HBM bytes of each main kernel, per step
        U, L = st["n_unique"], 300
        alg = {
            "k_msv": U * ((nprof + 63) // 64) * ((L + 15) // 16 * 4) + 2 * ((nprof + 63) // 64 * 64) * U,
            # per pair: packed read + PairRec/PairOut; per row: 6 special-state floats written
            "k_filters_fwd": st["n_past_msv"] * (((L + 15) // 16 * 4) + 16 + 40) + st["fwd_rows"] * 24,
            # per row: Forward's 6 floats read, 6 decoding terms written
            "k_bwd_decode": st["n_past_fwd"] * (((L + 15) // 16 * 4) + 16 + 40) + st["fwd_rows"] * 48,
            # per row: 5 terms read (nothing written back)
            "k_decode": st["n_past_fwd"] * (16 + 40) + st["fwd_rows"] * 20,
            # envelope sweeps: Backward rows written once, read once (26 float4 per row), + 88 B result per envelope
            "k_env_fwd+k_env_bwd+k_env_post": st["env_rows"] * 2 * 26 * 16 + st["n_domains"] * (16 + 88),
        }
        alg_bytes = alg[dom]
        # what bounds each of them: wave instructions per second against 1024 SIMDs x 2.4 GHz / 4 cycles (VALU-bound scans),
        # algorithmic GB/s against HBM for the streaming ones
        WAVE_INSTR_PEAK = 256 * 4 * 2.4e9 / 4
        vfrac = {
            "k_msv": (st["msv_cells"] / 20.5) / (kern["k_msv"] * 1e-3) / WAVE_INSTR_PEAK if kern["k_msv"] > 0 else None,        # 20.5 cells per wave instruction
            # VALU instructions per DP row counted in the ISA of the unrolled loops (instruction audit):
            # Forward 533, Backward 561 on the path without a rescale (480 of each are the recurrence's own packed mul/add)
            "k_filters_fwd": (st["fwd_rows"] / 64 * 533) / (kern["k_filters_fwd"] * 1e-3) / WAVE_INSTR_PEAK if kern["k_filters_fwd"] > 0 else None,
            "k_bwd_decode": (st["fwd_rows"] / 64 * 561) / (kern["k_bwd_decode"] * 1e-3) / WAVE_INSTR_PEAK if kern["k_bwd_decode"] > 0 else None,
        }
        kernel_table = {k: {"ms": round(kern[k], 3), "alg_GBps": round(alg[k] / (kern[k] * 1e-3) / 1e9, 1) if kern[k] > 0 else None,
                            "hbm_frac": round(alg[k] / (kern[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kern[k] > 0 else None,
                            "valu_issue_frac": round(vfrac[k], 3) if vfrac.get(k) else None} for k in kern}
        achieved = alg_bytes / (kern[dom] * 1e-3) / 1e9
        # launches of the dominant kernel in one step, for per-launch figures
        nl = {"k_msv": 1}.get(dom, max(1, int(st.get("n_batches", 1))))
        traffic = PMC_BYTES_PER_ROW[dom] * st["fwd_rows"] / nl if dom in PMC_BYTES_PER_ROW else None
        trimmed = int(((out[0][:, 0] >= 0) & (out[0][:, 1] >= 0) & (out[0][:, 0] < out[0][:, 1])).sum())
        res = {
            "metric": "reads/sec trimmed (ITS2, stand-in taxon Tracheophyta for Fungi)", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8+f32", "data": "synthetic",
            "config": {"workload": ("configs[1]: %d synthetic 300 bp single-end reads per GPU, ITS2, cluster_id=1.0 (pure derep)" % args.reads)
                       if (args.cluster_id >= 1.0 and args.taxa == "T") else
                       ("%d synthetic 300 bp single-end reads per GPU, --taxa All --region ITS2 (814 profiles, as configs[3]), cluster_id=1.0" % args.reads)
                       if args.cluster_id >= 1.0 else
                       ("%d synthetic 300 bp single-end reads per GPU, ITS2, cluster_id=%g (greedy clustering, row a2)" % (args.reads, args.cluster_id)),
                       "taxon": "Tracheophyta (stand-in: F.hmm absent from the reference mount)" if args.taxa == "T" else "All (every ITSx set in the mount; F.hmm absent)", "profiles": nprof,
                       "unique": int(st["n_unique"]), "pairs_past_msv": int(st["n_past_msv"]), "pairs_past_fwd": int(st["n_past_fwd"]),
                       "domains": int(st["n_domains"]), "reads_trimmed_rank0": trimmed, "parallelism": "reads sharded x%d%s" % (world, ", global derep" if args.global_derep else "")},
            "stage_ms": {k: round(v / K, 3) for k, v in acc.items()},
            "kernels": kernel_table,
            "concurrency": "k_bias of batch b+1 runs on a second stream beside k_decode of batch b: ms_bias_kernel is its stretched wall time, not extra step time",
            "cluster": None if args.cluster_id >= 1.0 else {"windows": int(st["cl_windows"]), "cut_windows": int(st["cl_cuts"]),
                                                            "alignments": int(st["cl_alignments"]), "centroids": int(st["n_unique"])},
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "launches_per_step": nl, "avg_launch_ms": kern[dom] / nl, "alg_bytes_per_launch": alg_bytes / nl,
                         "traffic": traffic,
                         "note": "achieved = algorithmic bytes / HIP-event time of the kernel; the dominant kernels are VALU-bound scans "
                                 "(SURVEY 8d), see 'valu'; traffic = PMC bytes per lane-row (profiles/round1_pmc_hbm_traffic_200k.md) x rows per launch"},
            "valu": {"msv_gcups": st["msv_cells"] / (kern["k_msv"] * 1e-3) / 1e9 if kern["k_msv"] > 0 else None,
                     "fwd_rows_per_s": st["fwd_rows"] / (kern["k_filters_fwd"] * 1e-3) if kern["k_filters_fwd"] > 0 else None,
                     "bwd_rows_per_s": st["fwd_rows"] / (kern["k_bwd_decode"] * 1e-3) if kern["k_bwd_decode"] > 0 else None,
                     "env_rows_per_s_x3_sweeps": 3 * st["env_rows"] / (kern["k_env_fwd+k_env_bwd+k_env_post"] * 1e-3) if kern["k_env_fwd+k_env_bwd+k_env_post"] > 0 else None,
                     "peak_lane_gops": VALU_PEAK_GOPS},
        }
        if args.cpu_sample != 0 and world == 1:          # the CPU baseline is a rank-0, N=1 leg only
            threads = os.cpu_count() or 1
            if args.cpu_sample < 0:                      # ~1 s per 80 reads per core on the scalar port
                args.cpu_sample = int(min(40000, max(1200, 120 * threads)))
            m = min(args.cpu_sample, args.reads)
            v, cdt, nc, ccoords = cpu_baseline(hmm, blob, offs, m, threads)
            # trim-coordinate concordance (BASELINE metric): the engine on the very same sample against the baseline path
            e2 = Engine(local_rank)
            e2.load_profiles(text=hmm)
            e2.set_reads_buffer(blob[:int(offs[m])], offs[:m + 1])
            e2.derep(strand_both=True, minseqlength=32)
            e2.search(T=10.0, F1=1e-6, F2=1e-6, F3=1e-6)
            e2.finalize(domE=10.0)
            gs, ge, gt, _ = e2.trim_coords("3_", "4_")
            e2.close()
            conc = float((np.stack([gs, ge, gt], axis=1) == ccoords).all(axis=1).mean())
            res["cpu_baseline"] = {"value": v, "unit": "reads/s", "cores": threads, "kind": "port",
                                   "sample": "first %d reads of the same workload (%d unique), derep+search+argmax, %.1f s" % (m, nc, cdt),
                                   "trim_coord_concordance": conc}
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
