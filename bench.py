#!/usr/bin/env python3
"""bench.py -- reads/sec trimmed on the hot path (BASELINE.json metric), 1..8 GPUs of one node.

Workload at N=1 (default): BASELINE.json configs[2] -- 10 M synthetic merged 2x300 bp reads (lengths 300-580, SURVEY 8d
cfg3), ITS2, cluster_id = 1.0; Fungi's model file is absent from the reference mount, so the stand-in taxon Tracheophyta
(155 ITS2 profiles) is used and labelled.  `--workload cfg1` is configs[1] (1 M x 300 bp single-end).

A "step" is one pass of the hot path over one batch of reads whose ASCII text is resident in HBM when the timed region
starts: 2-bit packing on the device -> derep -> MSV -> bias/Forward -> Backward/domain definition -> envelope re-scoring
-> [all-reduce domZ] -> thresholds/argmax -> per-read (start, stop, tlen) on the host of every rank -> gather to rank 0.
`host_handover` is the same step fed from a host buffer (staged PCIe upload included), measured in a leg of its own.

`--gpus N`: when N > 1 and no launcher set WORLD_SIZE, this process starts N ranks of itself (before anything touches the
GPU) and relays rank 0's line.  STRONG scaling by default (north_star: "10 M reads at 1 GPU, >= 6x further at 8 GPUs"):
the workload's reads (10 M for configs[2]) are sharded over the N ranks from ONE template library, so `--global-derep` has
cross-shard duplicates to find; `--total-reads T` sets another total; `--weak` gives every rank its own full-size shard.
`--workload cfg4` is BASELINE configs[4]'s per-GPU shape: 2x250-merged reads (300-480 bases), `--region ALL` profiles
(`1_` / `4_`), greedy clustering at 0.995 (row a2) instead of exact dereplication.

Prints ONE JSON line on rank 0.
"""
import argparse
import gzip
import json
import os
import shutil
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak (MI355X_MICROARCH.md)
# fp32 VALU ceiling WITHOUT fused multiply-add: 256 CUs x 4 SIMDs x 32 lanes (packed f32: 2 x 16) x 2.4 GHz = 78.6 TFLOP/s.
# HMMER rounds every product and every sum separately, so the DP kernels cannot use FMA (the 157 TFLOP/s figure).
VALU_NOFMA_TFLOPS = 256 * 4 * 32 * 2.4e9 / 1e12
WAVE_INSTR_PEAK = 256 * 4 * 2.4e9 / 4          # wave instructions per second: 1024 SIMDs, one VALU issue per 4 cycles
# the recurrence's own work per DP lane-row (45-node model, Q = 12 striped vectors of 4 floats), every product and every sum one flop:
#   Forward  (p7_ForwardParser): 16 vector operations per q (9 match, 1 + 2 + 1 delete, 3 insert) + two of the three lazy DD passes
#            (2 per q each) = 20 x 12 x 4 = 960 flops;
#   Backward (p7_BackwardParser): 3 (next row's M x emission, into B) + 7 (I, partial D, M) + 4 (first DD pass, E into D and M)
#            + 3 x 2 (its three further DD passes are unconditional) + 2 (D into M) = 22 per q = 22 x 12 x 4 = 1056 flops;
# the kernels issue 533 (Forward) / 561 (Backward) VALU instructions per row, 484 / 537 of them packed multiplies and adds
# (instruction audit; Backward's include the five decoding products of the row)
# k_fwd_bound (csrc/k_lazy.hip, the lazy stage's score-only Forward): nodes in their natural order, no striping -- per node 8 flops for the
#            match cell (4 products, 3 sums, the emission), 3 for the insert cell, 3 for the delete cell, 2 for the E sum = 16 x 46 nodes = 736
#            flops per lane-row; it is free to fuse (its result is a bound, not HMMER's float): 329 VALU instructions per row, 139 of them FMAs
FLOPS_PER_ROW = {"k_filters_fwd": 960.0, "k_bwd_decode": 1056.0, "k_fwd_bound": 736.0}     # k_fwd_bound: 16 x the models' mean node count at run time (720 for 45 nodes)
VALU_PER_ROW = {"k_filters_fwd": 533.0, "k_bwd_decode": 561.0, "k_fwd_bound": 329.0}
# Instructions per wave and DP row by issue class (scripts/isa_count.py, the row loop without its rare branches).  Round 5 measured what
# each class costs (profiles/round5_valu_issue.md, itsx_debug_issue): a PACKED instruction (v_pk_*: two f32 or two 16-bit cells per lane)
# holds a SIMD 1.74 ns, a plain VALU instruction 1.00 ns, an s_nop 0.4 ns at full occupancy -- rounds 1-4 priced every instruction at
# 4 cycles of a 2.4 GHz clock (1.67 ns), which read 1.02 for k_msv.  valu_issue_frac = sum(count x ns) x wave-rows / (1024 SIMDs x time).
ISSUE_MIX = {"k_msv": {"packed": 69, "plain": 19, "s_nop": 3},
             "k_fwd_bound": {"packed": 185, "plain": 77, "s_nop": 6},
             "k_filters_fwd": {"packed": 484, "plain": 49, "s_nop": 21},
             "k_bwd_decode": {"packed": 537, "plain": 24, "s_nop": 40}}
ISSUE_NS = {"packed": 1.74, "plain": 1.00, "s_nop": 0.40}
ISSUE_SOURCE = "built-in (profiles/round5_valu_issue.json not found)"
N_SIMD = 1024


def _load_issue():
    global ISSUE_NS, ISSUE_SOURCE
    p = os.path.join(ROOT, "profiles", "round5_valu_issue.json")
    if os.path.exists(p):
        with open(p) as f:
            d = json.load(f)
        ISSUE_NS = {k: float(d["ns_per_simd"][k]) for k in ("packed", "plain", "s_nop")}
        ISSUE_SOURCE = "profiles/round5_valu_issue.json"


def issue_frac(kernel, wave_rows, ms):
    """share of the SIMDs' issue time the kernel's own instruction mix accounts for (1.0 = nothing but its instructions, back to back)"""
    if not ms or ms <= 0:
        return None
    ns_row = sum(ISSUE_MIX[kernel][c] * ISSUE_NS[c] for c in ISSUE_NS)
    return wave_rows * ns_row / (N_SIMD * ms * 1e6)
USES_FMA = {"k_fwd_bound"}       # priced against the FMA peak; the HMMER-order kernels against the no-FMA ceiling
VALU_FMA_TFLOPS = 157.3          # the guide's fp32 vector peak (every instruction a fused multiply-add)
# HBM bytes per lane-row from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE corrected by the factor
# calibrated with scripts/fetch_calib.py on the slab access pattern (profiles/round2_fetch_calibration.md)
PMC_BYTES_PER_ROW = None         # filled from profiles/round4_pmc_bytes_per_row.json (this round's kernels) when present
PMC_SOURCE = None


def _load_pmc():
    global PMC_BYTES_PER_ROW, PMC_SOURCE
    for name in ("round6_pmc_bytes_per_row.json", "round5_pmc_bytes_per_row.json", "round4_pmc_bytes_per_row.json", "round2_pmc_bytes_per_row.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            with open(p) as f:
                PMC_BYTES_PER_ROW = json.load(f)
            PMC_SOURCE = "profiles/" + name
            return


CFG3_SHARE = 6250000             # configs[3]: 50 M reads, --taxa All (814 ITS2 profiles), read-sharded over 8 GPUs = 6.25 M reads per GPU
CFG4_DEFAULT = 2000000           # reads of a default `--workload cfg4` run (configs[4]'s per-GPU share is 12.5 M: --reads 12500000)


def region_profiles(hmm_text, left="3_", right="4_"):
    """what create_runtime_hmm selects (main.py:200-208): the profile blocks whose NAME starts with one of the region's prefixes"""
    blocks = [b + "//\n" for b in hmm_text.split("//\n") if "NAME  " in b]
    return "".join(b for b in blocks if b.split("NAME  ")[1][:2] in (left, right))


def its2_profiles(hmm_text):
    return region_profiles(hmm_text, "3_", "4_")


def launch_ranks(n):
    """N > 1 and no launcher: start N ranks of this script (nothing here has touched the GPU), relay rank 0's output."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def shard_plan(workload, reads, total_reads, weak, world, rank):
    """How the workload's reads are laid over the ranks.  N > 1 defaults to STRONG scaling: the workload's own size (10 M reads
    for configs[2]) is the total of the whole job and rank r takes the slice [T r / N, T (r + 1) / N) -- north_star's ">= 6x
    further at 8 GPUs" is on the 10 M-read job.  --weak (or an explicit --reads) gives every rank a full-size shard of its own.
    Returns (reads of this rank, total of the job or 0 when weak, "strong" | "weak")."""
    own = {"cfg2": 10000000, "cfg1": 1000000, "cfg4": CFG4_DEFAULT, "cfg3": CFG3_SHARE}[workload]
    if world > 1 and not weak and not reads and not total_reads and workload != "cfg3":      # (cfg3 is 50 M reads over 8 GPUs: each rank its share)
        total_reads = own
    if total_reads > 0:
        return total_reads * (rank + 1) // world - total_reads * rank // world, total_reads, "strong"
    return reads or own, 0, "weak"


def launch_selftest(world, rank, args):
    """CPU check of the launcher + the sharding plan + the two exchange steps with gloo and no engine (tests/test_dist_gloo.py)."""
    import numpy as np
    import torch.distributed as dist
    from itsxpress_amd.dist import allreduce_domz, gather_coords
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local, total, scaling = shard_plan(args.workload, args.reads, args.total_reads, args.weak, world, rank)
    plan = allreduce_domz(np.array([n_local], np.int64))
    z = allreduce_domz(np.full(5, rank + 1, np.int64))
    n = 3 + rank
    c = gather_coords(np.full(n, rank, np.int32), np.arange(n, dtype=np.int32), np.full(n, 7, np.int32), np.ones(n, np.int32))
    dist.barrier()
    if rank == 0:
        ok = bool((z == world * (world + 1) // 2).all()) and [b.shape[0] for b in c] == [3 + r for r in range(world)] and \
            all(int(b[0, 0]) == r for r, b in enumerate(c))
        os.write(REAL_STDOUT, (json.dumps({"metric": "launcher self-test (no engine, gloo)", "value": None, "n_gpus": world, "ok": ok, "scaling": scaling,
                          "reads_rank0": int(n_local), "reads_all_ranks": int(plan[0]), "total_reads": int(total)}) + "\n").encode())
    dist.destroy_process_group()


def real_tools_baseline(hmm_its2, seqs, threads, tmp):
    """SURVEY 8d's preferred baseline: the reference's own command lines (itsxpress/SeqSample.py:106-116, 191-209) when
    vsearch and hmmsearch are on PATH.  Returns (seconds, per-read coordinates) or None."""
    if shutil.which("vsearch") is None or shutil.which("hmmsearch") is None:
        return None
    import numpy as np
    from itsxpress_amd.SeqSample import Dedup, ItsPosition
    fq, hmm = os.path.join(tmp, "seq.fq"), os.path.join(tmp, "runtime.hmm")
    with open(fq, "w") as f:
        for i, s in enumerate(seqs):
            f.write("@r%09d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    with open(hmm, "w") as f:
        f.write(hmm_its2)
    uc, rep, dom = (os.path.join(tmp, n) for n in ("uc.txt", "rep.fa", "domtbl.txt"))
    t0 = time.time()
    subprocess.run(["vsearch", "--fastx_uniques", fq, "--fastaout", rep, "--uc", uc, "--strand", "both"], check=True,
                   stderr=subprocess.DEVNULL)
    subprocess.run(["hmmsearch", "--domtblout", dom, "-T", "10", "--cpu", str(threads), "--tformat", "fasta", "--F1", "1e-6",
                    "--F2", "1e-6", "--F3", "1e-6", hmm, rep], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dd = Dedup(uc_file=uc, rep_file=rep, seq_file=fq)
    pos = ItsPosition(domtable=dom, region="ITS2")
    dt = time.time() - t0
    coords = np.full((len(seqs), 3), -1, np.int32)
    for i in range(len(seqs)):
        r = dd.matchdict.get("r%09d" % i)
        if r is None:
            continue
        try:
            a, b, c = pos.get_position(r)
        except KeyError:
            continue
        coords[i] = [-1 if a is None else a, -1 if b is None else b, -1 if c is None else c]
    return dt, coords


def cpu_baseline(hmm_text, blob, offs, sample_reads, threads, lp="3_", rp="4_", cluster_id=1.0):
    """the oracle (a port of the reference's CPU path) timed on a bounded sample of the same workload"""
    import numpy as np
    import orc
    # the timed leg runs oracle/libbase_sse.so: the checker's sources with HMMER's 4-lane float vectors in real SSE2 registers
    # and a 16-lane byte MSV filter (results bit-identical to the scalar checker: tests/test_oracle_cpu.py)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libbase_sse.so"], check=True)
    orc.use_library("libbase_sse.so")
    raw = bytes(blob[:int(offs[sample_reads])])
    seqs = [raw[offs[i]:offs[i + 1]].decode() for i in range(sample_reads)]
    hs = orc.HmmSet(text=hmm_text)
    t0 = time.time()
    codes, o = orc.digitize(seqs)
    if cluster_id < 1.0:            # row a2: the sequential greedy procedure (one thread, like its definition)
        cl = orc.cluster(codes, o, ["r%09d" % i for i in range(len(seqs))], cluster_id)
        nc, rep = cl["n_centroids"], cl["rep_of"]
    else:
        nc, rep, strand = orc.derep(codes, o)
    seeds = [i for i in range(len(seqs)) if rep[i] == i]
    c2, o2 = orc.digitize([seqs[i] for i in seeds])
    res = orc.SearchResult(hs, c2, o2, threads=threads, keep_trace=0)
    us, ue, ut, ui = res.positions(lp, rp)
    dt = time.time() - t0
    rep = np.asarray(rep)
    uniq = np.cumsum(rep == np.arange(len(seqs))) - 1          # unique index of each seed, in input order
    uo = uniq[np.maximum(rep, 0)]
    coords = np.stack([us[uo], ue[uo], ut[uo]], axis=1)                      # per read of the sample: the baseline path's answer
    coords[rep < 0] = -1                                                     # reads the grouping dropped (--minseqlength)
    cnt = res.counts
    ulen = np.array([len(seqs[i]) for i in seeds], np.float64)
    # DP cells of the float stages actually run (Forward on every pair past the bias filter, Backward on every pair past Forward;
    # a 45-node model: 45 cells per row), per second and per THREAD -- the figure to hold against HMMER's published rates
    dp_cells = float(ulen.mean()) * 45.0 * (cnt.get("past_bias", 0) + cnt.get("past_fwd", 0))
    orc.use_library("liborc.so")
    extra = {"threads": threads, "physical_cores": physical_cores(), "simd": "SSE2 (4 x f32 Forward/Backward, 16 x u8 MSV), bit-identical to the scalar checker",
             "fwd_bwd_cells_per_s_per_thread": dp_cells / dt / max(threads, 1), "msv_cells_per_s_per_thread": float(ulen.sum()) * 45.0 * hs.n / dt / max(threads, 1),
             "note": "per-thread rates are whole-leg averages (every stage's time included), so they understate each kernel's own rate"}
    return sample_reads / dt, dt, nc, coords, seqs, extra


def stage_a(acc, K, n_reads, mean_len, cluster_id):
    ms_pack, ms_group = acc.get("ms_pack", 0.0) / K, (acc.get("ms_cluster", 0.0) if cluster_id < 1.0 else acc.get("ms_derep", 0.0)) / K
    alg = n_reads * (-(-mean_len // 4) + 48.0)
    text = n_reads * mean_len
    out = {"ms_pack": round(ms_pack, 3), "ms_group": round(ms_group, 3), "alg_bytes": alg, "text_bytes": text,
           "note": "pack = ASCII text -> 2-bit words + exception list (reads the text once, writes L / 4); group = hash, table insert, exact "
                   "verification, unique lists (row a1) or the greedy clustering (row a2: alignment-bound, not a streaming stage)"}
    if ms_pack > 0:
        out["pack_GBps"] = round((text + n_reads * mean_len / 4.0) / (ms_pack * 1e-3) / 1e9, 1)
        out["pack_hbm_frac"] = round(out["pack_GBps"] / HBM_PEAK_GBS, 4)
    if ms_group > 0 and cluster_id >= 1.0:
        out["derep_GBps_on_alg_bytes"] = round(alg / (ms_group * 1e-3) / 1e9, 1)
        out["derep_hbm_frac_on_alg_bytes"] = round(out["derep_GBps_on_alg_bytes"] / HBM_PEAK_GBS, 4)
    return out


def effective_cpus():
    """CPUs this process may actually use: the scheduler affinity mask cut by the cgroup's CPU quota (a GPU box hands a 1-GPU job
    16 CPUs' worth of a 256-thread host: starting 256 threads there only adds throttling).  Returns (usable CPUs, what limits them)."""
    n, why = os.cpu_count() or 1, "os.cpu_count()"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, why = a, "sched_getaffinity"
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = int(float(quota) / period + 0.5)
                if 0 < q < n:
                    n, why = q, "cgroup CPU quota (%s)" % path
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n), why


def physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo: os.cpu_count() counts hardware THREADS"""
    try:
        seen, phys = set(), None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                seen.add((phys, ln.split(":")[1].strip()))
        return len(seen) or None
    except OSError:
        return None


T_START = time.time()
REAL_STDOUT = 1


def progress(msg):
    """a line on stderr now and then: a run that is silent for minutes looks hung to the harness (stdout carries the ONE JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %6.1fs] %s" % (time.time() - T_START, msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["cfg2", "cfg1", "cfg3", "cfg4"], default="cfg2",
                    help="cfg2 = BASELINE configs[2] (10 M merged reads, 300-580 bases; the default), cfg1 = configs[1] (1 M x 300), "
                         "cfg3 = configs[3]'s per-GPU share (6.25 M merged reads against --taxa All: 814 ITS2 profiles; every rank its own share), "
                         "cfg4 = configs[4]'s shape (2x250-merged reads of 300-480 bases, --region ALL profiles, cluster_id 0.995)")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the workload's own size); implies --weak at N > 1")
    ap.add_argument("--total-reads", type=int, default=0,
                    help="strong scaling: this many reads in total, sharded over the ranks (shards share their templates); "
                         "the default at N > 1, with the workload's own size as the total")
    ap.add_argument("--weak", action="store_true", help="N > 1: every rank its own shard of the workload's size (weak scaling)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads in the CPU-baseline sample (0 = skip, -1 = auto: ~20-30 s of CPU work)")
    ap.add_argument("--handover-steps", type=int, default=1, help="steps fed from the host buffer, outside `value` (0 = skip)")
    ap.add_argument("--paired-pairs", type=int, default=0,
                    help="N = 1 only: after the timed steps, configs[2] as the config states it -- this many synthetic 2x250 PAIRS, R1 / R2 .fastq.gz in, "
                         "trimmed R1 / R2 .fastq.gz out, through the mirror classes in arrays mode (scripts/paired_run.py) -- reported as the extra key "
                         "`paired_file_to_file`, never as `value` (0 = skip, the default: generating 10 M pairs alone takes minutes)")
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("ITSX_BENCH_BUDGET_S", "520")),
                    help="wall-clock budget of the whole run: the legs AFTER the K timed steps (host hand-over, CPU baseline) shrink or "
                         "are skipped to stay inside it (the driver's limit is 600 s); the timed steps themselves are never cut")
    ap.add_argument("--cluster-id", type=float, default=None,
                    help="1.0 = exact dereplication (the default for cfg1 / cfg2); < 1 runs row a2 (greedy clustering) instead (cfg4: 0.995)")
    ap.add_argument("--taxa", choices=["T", "all"], default="T",
                    help="T = the stand-in taxon (155 ITS2 profiles); all = --taxa All --region ITS2 (814 profiles, configs[3])")
    ap.add_argument("--global-derep", action="store_true", help="(the default at N > 1 since round 4; kept for old command lines)")
    ap.add_argument("--per-shard-derep", action="store_true",
                    help="N > 1: every rank dereplicates and scores its own shard only (SURVEY 8e option 1): a sequence present in two shards is "
                         "scored twice and counted twice in domZ, so coordinates can differ from the one-GPU answer near the thresholds.  The "
                         "default is EXACT global dereplication (option 2): the result of one GPU on the whole input, read for read")
    ap.add_argument("--concordance-reads", type=int, default=12000000,
                    help="N > 1: rank 0 re-runs the whole job on ONE engine after the timed steps and compares every coordinate for equality "
                         "(`concordance_vs_single_engine`) when the job has at most this many reads (0 = skip)")
    ap.add_argument("--rows", choices=["lazy", "compact", "full"], default="lazy",
                    help="what the search keeps of the domain table (itsx_set_rows_mode): lazy = pairs that cannot win ItsPosition's argmax "
                         "are not evaluated past their Forward score (the default: the step asks for coordinates, not for domtbl.txt); compact = "
                         "every pair evaluated, only the rows that can still win kept; full = every row resident (66 GB at 10 M reads)")
    ap.add_argument("--full-rows", action="store_true", help="= --rows full")
    ap.add_argument("--alone-steps", type=int, default=1, help="1: one extra step with ITSX_MSV_OVERLAP=0 for the kernels' alone times (0 = skip)")
    ap.add_argument("--full-steps", type=int, default=1,
                    help="with --rows lazy: this many extra steps with every pair evaluated (--rows compact), after the timed ones and outside "
                         "`value`: `full_pipeline_value`, and the lazy coordinates are compared with them for equality (0 = skip)")
    ap.add_argument("--files-leg", type=int, default=1,
                    help="N = 1, --rows lazy: after the timed steps the job's uc.txt, rep.fa and the winners' domtbl.txt are written once (what "
                         "ITSXPRESS_DOMTBL=winners hands the reference's own parsers), outside `value`: `winners_files` (0 = skip)")
    ap.add_argument("--launch-selftest", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    # stdout carries ONE JSON line and nothing else: whatever a library prints there (RCCL's version banner at the first collective) goes
    # to stderr; the line is written to the real stdout at the very end
    global REAL_STDOUT
    sys.stdout.flush()
    REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    if args.cluster_id is None:
        args.cluster_id = 0.995 if args.workload == "cfg4" else 1.0
    # N > 1: exact global dereplication unless asked otherwise (greedy clustering has no exact sharded form: per shard)
    args.global_derep = (not args.per_shard_derep) and args.cluster_id >= 1.0

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_selftest:
        return launch_selftest(world, rank, args)
    import numpy as np
    import torch
    import torch.distributed as dist
    if os.environ.get("ITSX_BENCH_ONE_GPU") == "1":      # rehearsal of the N-rank code path on a one-GPU box (with ITSX_BENCH_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("ITSX_FORCE_DIST") == "1"     # the latter: exercise RCCL with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("ITSX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = dev if os.environ.get("ITSX_BENCH_BACKEND", "nccl") == "nccl" else torch.device("cpu")     # where the bench's own scalars are reduced

    if args.full_rows:
        args.rows = "full"
    from itsxpress_amd import Engine
    from itsxpress_amd.dist import agree, exchange_and_finalize, exchange_rows, gather_rows, global_derep, read_rows
    import synth
    _load_pmc()
    _load_issue()

    from itsxpress_amd.definitions import hmm_path
    fungi = hmm_path("Fungi") if args.taxa == "T" else None      # ITSx_db/HMMs/F.hmm via $ITSXPRESS_DB_DIR / an installed itsxpress
    if fungi:                                                     # the taxon every BASELINE config names: used the moment it is supplied
        with open(fungi) as f:
            thmm = f.read()
        progress("Fungi profiles from %s" % fungi)
    else:
        with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f:
            thmm = f.read()
    cfg3 = args.workload == "cfg3"
    if cfg3:
        args.taxa = "all"
    cfg2, cfg4 = args.workload == "cfg2" or cfg3, args.workload == "cfg4"        # (cfg3 has cfg2's read shape)
    lp, rp = ("1_", "4_") if cfg4 else ("3_", "4_")        # create_runtime_hmm's prefixes: --region ALL / ITS2 (main.py:200-208)
    hmm = region_profiles(thmm, lp, rp)
    if args.taxa == "all":
        if cfg4:
            raise SystemExit("--taxa all is configs[3]'s profile set (ITS2); cfg4 runs the stand-in taxon's --region ALL profiles")
        with gzip.open(os.path.join(ROOT, "tests", "golden", "all_its2.hmm.gz"), "rt") as f:
            hmm = f.read()
    # the workload's own size per GPU: configs[2] 10 M; configs[1] 1 M; configs[4] 100 M over 8 GPUs = 12.5 M per GPU (the
    # default runs CFG4_DEFAULT reads so that a plain `bench.py --workload cfg4` finishes in minutes; --reads 12500000 is the share)
    n_local, args.total_reads, scaling = shard_plan(args.workload, args.reads, args.total_reads, args.weak, world, rank)
    strong = scaling == "strong"
    progress("generating %d reads (%s)" % (n_local, args.workload))
    t_gen = time.time()
    cfgno = 4 if cfg3 else (3 if cfg2 else (5 if cfg4 else 2))
    gen = dict(config=cfgno, seed=synth.SEED + cfgno + 1000 * rank, as_array=True, left=lp, right=rp)
    if cfg2:
        gen.update(fixed_len=0, len_range=(300, 580))
    if cfg4:
        gen.update(fixed_len=0, len_range=(300, 480))
    if strong:                      # one template library for the whole job: 2 % of the TOTAL reads
        gen.update(template_seed=synth.SEED + 77, frac_templates=0.02 * args.total_reads / max(n_local, 1))
    blob, offs = synth.make_reads(thmm, n_local, **gen)
    t_gen = time.time() - t_gen
    mean_len = float(offs[-1]) / max(n_local, 1)
    eng = Engine(local_rank)
    eng.set_rows_mode(args.rows)
    nprof = eng.load_profiles(text=hmm)
    # the models' node counts (LENG lines of the HMMER3/f text): k_fwd_bound's algorithmic flops are 16 per node and lane-row
    lengs = [int(ln.split()[1]) for ln in hmm.split("\n") if ln.startswith("LENG ")]
    mean_nodes = float(sum(lengs)) / max(len(lengs), 1) if lengs else 45.0
    d_blob = torch.from_numpy(blob).to(dev)      # the batch's ASCII text, resident in HBM before the timed region
    torch.cuda.synchronize()

    comm = {"allreduce_ms": 0.0, "gather_ms": 0.0, "lazy_reruns": 0}

    def guarded(fn):
        """an engine call of this rank; at N > 1 every rank then learns whether all of them got through (a failed rank must not
        leave its peers waiting in the next collective: they all stop with its message)"""
        err = None
        try:
            fn()
        except Exception as e:                       # noqa: reported below, on every rank
            err = e
        if use_dist and not agree(err is None, cdev):
            raise RuntimeError("rank %d: %s" % (rank, repr(err) if err is not None else "another rank's engine call failed"))
        if err is not None:
            raise err

    calls = {}                                        # wall time of each engine call of a step, as the host sees it (config.host_calls_ms)

    def timed_call(name, fn):
        t = time.perf_counter()
        r = fn()
        calls[name] = calls.get(name, 0.0) + (time.perf_counter() - t) * 1e3
        return r

    def step(from_host=False):
        def group():
            if from_host:
                timed_call("set_reads", lambda: eng.set_reads_buffer(blob, offs))     # staged upload + device packing
            else:
                timed_call("set_reads", lambda: eng.set_reads_device(d_blob.data_ptr(), offs, keep=d_blob))     # device packing of the resident text
            if args.cluster_id < 1.0:
                timed_call("cluster", lambda: eng.cluster(args.cluster_id, strand_both=True))
            else:
                timed_call("derep", lambda: eng.derep(strand_both=True, minseqlength=1))
        guarded(group)
        g = global_derep(eng, n_local, cdev) if (use_dist and args.global_derep) else None
        search = lambda: timed_call("search", lambda: eng.search(T=10.0, F1=1e-6, F2=1e-6, F3=1e-6))
        guarded(search)
        if not use_dist:
            timed_call("finalize", lambda: eng.finalize(domE=10.0))                   # (a lazy search with undecided rows repeats itself in full in here)
            return [timed_call("trim_coords", lambda: eng.trim_coords(lp, rp))]          # (start, stop, tlen, index) per read, on the host
        # N > 1: the two exchanges run on the engine's own device buffers (RCCL over xGMI), nothing bounces through numpy
        tc = time.perf_counter()
        # hmmsearch's domZ is a count over the WHOLE data set (after a lazy search: its bounds); thresholds; a full search on
        # every rank if some rank's rows stay undecided
        comm["lazy_reruns"] += 1 if exchange_and_finalize(eng, dev, 10.0, search) > 0 else 0
        comm["allreduce_ms"] += (time.perf_counter() - tc) * 1e3
        if g is not None:            # coordinates of the uniques scored elsewhere arrive here, then fan out to the reads
            rows = read_rows(eng, exchange_rows(g, eng.rep_coords_device(lp, rp, dev)), dev)
        else:
            rows = eng.trim_coords_device(lp, rp, dev)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        out = gather_rows(rows, dst=0)                 # rank 0 ends with the rows on its host; includes the wait for the slowest rank
        torch.cuda.synchronize()
        comm["gather_ms"] += (time.perf_counter() - tc) * 1e3
        return out

    progress("reads generated in %.1f s, text resident in HBM; %d warm-up + %d timed steps" % (t_gen, args.warmup, args.steps))
    warm_done = 0
    for k in range(args.warmup):
        tw = time.perf_counter()
        step()
        tw = time.perf_counter() - tw
        warm_done += 1
        progress("warm-up step %d/%d done (%.2f s)" % (k + 1, args.warmup, tw))
        if k >= 1 and k + 1 < args.warmup:
            # The K timed steps are never cut.  Warm-up steps after the second are (the first one pays for every first-use
            # allocation and says little about a step): when another one would push the K timed steps past --budget-s (a slow
            # box, a cold page cache, N ranks on one host), warming up stops here and the line reports the number actually
            # done ("warmup") beside the number asked for ("warmup_requested").
            room = torch.tensor([args.budget_s - (time.time() - T_START) - tw - (args.steps * tw * 1.05 + 20.0)], dtype=torch.float64, device=cdev)
            if use_dist:
                dist.all_reduce(room, op=dist.ReduceOp.MIN)              # every rank takes the same decision
            if float(room.item()) < 0:
                progress("warm-up cut at %d of %d steps: the %d timed steps need the rest of the %.0f-s budget" % (warm_done, args.warmup, args.steps, args.budget_s))
                break
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    comm["allreduce_ms"] = comm["gather_ms"] = 0.0
    calls.clear()
    t0 = time.perf_counter()
    acc = {}
    out = None
    for k in range(args.steps):
        out = step()
        progress("timed step %d/%d done (%.2f s so far)" % (k + 1, args.steps, time.perf_counter() - t0))
        st = eng.stats()
        for k, v in st.items():
            if k.startswith("ms_"):
                acc[k] = acc.get(k, 0.0) + v
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0               # this rank's own K steps (the gather's wait for slower ranks included)
    host_calls_ms = {k: round(v / max(args.steps, 1), 1) for k, v in calls.items()}     # (before the extra legs below add to `calls`)
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    rank_ms = None
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([n_local], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        total_local = int(tot.item())
        # per-rank figures: step time without the exchanges (what the rank computed), and the exchanges themselves
        own = (dt_own * 1e3 - comm["allreduce_ms"] - comm["gather_ms"]) / max(args.steps, 1)
        v = torch.tensor([own, -own, comm["allreduce_ms"] / max(args.steps, 1), comm["gather_ms"] / max(args.steps, 1)], dtype=torch.float64, device=cdev)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        rank_ms = {"compute_ms_per_step_max": float(v[0]), "compute_ms_per_step_min": -float(v[1]),
                   "allreduce_ms_per_step_max": float(v[2]), "gather_ms_per_step_max": float(v[3]),
                   "note": "per rank: step time without the two exchanges (max / min over ranks), the domZ all-reduce, the coordinate gather "
                           "(a rank's gather includes its wait for the slowest rank)"}
    else:
        total_local = n_local
    st = eng.stats()

    # one more step with nothing beside the scan kernels (ITSX_MSV_OVERLAP=0: the next chunk's k_msv does not run beside the domain stage and
    # the bound pass), outside `value`: the kernels' ALONE times, which the roofline block uses when the timed steps' are stretched
    alone = None
    if args.alone_steps > 0 and args.cluster_id >= 1.0 and args.budget_s - (time.time() - T_START) > 2.5 * dt / max(args.steps, 1) + 30.0:
        os.environ["ITSX_MSV_OVERLAP"] = "0"
        progress("step without the MSV overlap (kernels alone)")
        step()
        torch.cuda.synchronize()
        del os.environ["ITSX_MSV_OVERLAP"]
        sa = eng.stats()
        alone = {"k_msv": sa["ms_msv_kernel"], "k_fwd_bound": sa["ms_bound_kernel"] - sa.get("ms_bwd_bound", 0.0), "k_bwd_bound": sa.get("ms_bwd_bound", 0.0), "k_filters_fwd": sa["ms_fwd_kernel"], "k_bwd_decode": sa["ms_bwd_kernel"],
                 "k_decode": sa["ms_decode_kernel"], "k_env_fwd+k_env_bwd+k_env_post": sa["ms_env_kernel"]}
        if use_dist:
            dist.barrier()

    # the same step with EVERY pair evaluated (--rows compact), outside `value`: what the lazy stage saves, and that it changes nothing
    full_leg = None
    if args.rows == "lazy" and args.full_steps > 0 and args.cluster_id >= 1.0:
        left = torch.tensor([args.budget_s - (time.time() - T_START)], dtype=torch.float64, device=cdev)
        if use_dist:
            dist.all_reduce(left, op=dist.ReduceOp.MIN)
        if float(left.item()) < args.full_steps * 3.0 * dt / max(args.steps, 1) + 30.0:
            full_leg = {"skipped": "wall-clock budget (--budget-s %.0f)" % args.budget_s}
        else:
            lazy_rows = np.stack(out[0], axis=1) if (out is not None and isinstance(out[0], tuple)) else (np.concatenate(out) if out is not None else None)
            eng.set_rows_mode("compact")
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            if float(left.item()) > (args.full_steps + 1) * 3.0 * dt / max(args.steps, 1) + 60.0:
                progress("full-pipeline warm-up step")
                step()                                # (its work buffers are several times the lazy stage's: allocated here, not in the timed step)
                torch.cuda.synchronize()
            tf0 = time.perf_counter()
            fout = None
            for _ in range(args.full_steps):
                progress("full-pipeline step (every pair evaluated)")
                fout = step()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            tfull = time.perf_counter() - tf0
            stf = eng.stats()
            eng.set_rows_mode(args.rows)
            if use_dist:
                t = torch.tensor([tfull], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                tfull = float(t.item())
            full_leg = {"steps": args.full_steps, "ms_per_step": tfull / args.full_steps * 1e3, "pairs_past_fwd": int(stf["n_past_fwd"]), "domains": int(stf["n_domains"])}
            if rank == 0 and fout is not None:
                full_rows = np.stack(fout[0], axis=1) if isinstance(fout[0], tuple) else np.concatenate(fout)
                full_leg["coordinates_equal_lazy"] = bool(lazy_rows.shape == full_rows.shape and np.array_equal(lazy_rows, full_rows))

    # the same step fed from the host buffer (PCIe-inclusive), never part of `value`
    handover = None
    step_s = dt / max(args.steps, 1)
    left = torch.tensor([args.budget_s - (time.time() - T_START)], dtype=torch.float64, device=cdev)
    if use_dist:
        dist.all_reduce(left, op=dist.ReduceOp.MIN)                      # every rank takes the same decision
    left = float(left.item())
    cpu_leg_s = 0.0 if (args.cpu_sample == 0 or world > 1) else 40.0
    if args.handover_steps > 0 and left < args.handover_steps * step_s * 1.15 + cpu_leg_s + 15.0:
        args.handover_steps = 0
        handover = {"skipped": "wall-clock budget (--budget-s %.0f): %.0f s left after the timed steps" % (args.budget_s, left)}
    if args.handover_steps > 0:
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        th = time.perf_counter()
        mp = 0.0
        for _ in range(args.handover_steps):
            progress("host hand-over step")
            step(from_host=True)
            mp += eng.stats()["ms_pack"]
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        th = time.perf_counter() - th
        if use_dist:
            t = torch.tensor([th], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            th = float(t.item())
        handover = {"value": total_local * args.handover_steps / th, "unit": "reads/s", "steps": args.handover_steps,
                    "ms_per_step": th / args.handover_steps * 1e3, "ms_upload_and_pack": mp / args.handover_steps,
                    "GB_per_step": float(offs[-1]) / 1e9,
                    "note": "host buffer -> pinned staging -> HBM -> device packing inside the step (PCIe-inclusive); not `value`"}

    # the file-compatible outputs of the lazy search (ITSXPRESS_DOMTBL=winners, INTEGRATION.md 3a'): the three files the reference's own
    # Dedup / ItsPosition parse, written from the engine's state after the last step; never part of `value`
    files_leg = None
    if args.files_leg > 0 and world == 1 and args.rows == "lazy" and args.cluster_id >= 1.0:
        if args.budget_s - (time.time() - T_START) < (40.0 if args.cpu_sample == 0 else 80.0):
            files_leg = {"skipped": "wall-clock budget (--budget-s %.0f)" % args.budget_s}
        else:
            import tempfile
            progress("uc.txt, rep.fa and the winners' domtbl.txt of the job")
            if eng.stats()["lazy"] != 1:                  # (the last extra step was the full-pipeline one: the files come from a lazy search)
                step()
            fdir = tempfile.mkdtemp(prefix="itsx_bench_files_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
            try:
                tfw = time.perf_counter()
                eng.set_kept_rows(True)
                paths = [os.path.join(fdir, nm) for nm in ("uc.txt", "rep.fa", "domtbl.txt")]
                eng.write_uc(paths[0])
                eng.write_rep_fasta(paths[1])
                t_dom = time.perf_counter()
                eng.write_domtbl(paths[2])
                tfw, t_dom = time.perf_counter() - tfw, time.perf_counter() - t_dom
                files_leg = {"s_three_files": round(tfw, 3), "s_domtbl": round(t_dom, 3), "domtbl_rows": int(eng.L.itsx_num_domains(eng.h)),
                             "MB": {os.path.basename(q): round(os.path.getsize(q) / 1e6, 1) for q in paths},
                             "reads_per_s_search_plus_files": total_local / (dt / max(args.steps, 1) + tfw),
                             "note": "one row per target and profile prefix: the row ItsPosition.parse ends up with (tests/test_gpu_winners.py); "
                                     "written to a memory-backed directory; the default file mode lists every row (full_pipeline + ~136 rows per "
                                     "representative: profiles/round6_winners_run_10M.json)"}
            finally:
                eng.set_kept_rows(False)
                shutil.rmtree(fdir, ignore_errors=True)

    paired_leg = None
    if args.paired_pairs > 0 and world == 1:
        progress("paired file-to-file leg (%d pairs)" % args.paired_pairs)
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        keep = {k: os.environ.get(k) for k in ("ITSXPRESS_ARRAYS", "ITSXPRESS_STREAM")}
        os.environ["ITSXPRESS_ARRAYS"], os.environ["ITSXPRESS_STREAM"] = "1", "0"
        try:
            import paired_run
            paired_leg = paired_run.run(args.paired_pairs, True, stream=True)
            paired_leg["note"] = ("R1 / R2 .fastq.gz -> merge (k_merge.hip) -> derep -> lazy search -> coordinates -> trimmed R1 / R2 .fastq.gz, one GPU, "
                                  "ITSXPRESS_ARRAYS=1 (scripts/paired_run.py); host codecs included; not `value`.  `streamed` (round 6): the same files through "
                                  "the streamed paired pipeline -- R1 / R2 inflated side by side, merged chunk by chunk, the two outputs deflated while later "
                                  "chunks are scored (ITSXPRESS_STREAM=1, SeqSample.plan_output_paired)")
        finally:
            for k, v in keep.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    if rank == 0:
        total_reads = total_local * args.steps
        value = total_reads / dt
        K = args.steps
        # (two-sided sharing, round 6: ms_bound_kernel is pass A's wall time; ms_bwd_bound the share of its Backward chains, k_bwd_bound)
        kern = {"k_msv": acc["ms_msv_kernel"] / K, "k_fwd_bound": (acc.get("ms_bound_kernel", 0.0) - acc.get("ms_bwd_bound", 0.0)) / K, "k_filters_fwd": acc["ms_fwd_kernel"] / K,
                "k_bwd_decode": acc["ms_bwd_kernel"] / K, "k_decode": acc["ms_decode_kernel"] / K,
                "k_env_fwd+k_env_bwd+k_env_post": acc["ms_env_kernel"] / K, "k_bwd_bound": acc.get("ms_bwd_bound", 0.0) / K}
        dom = max(kern, key=kern.get)
        # algorithmic HBM bytes of each main kernel, per step; L = the mean read length
        U, wbytes = st["n_unique"], 4.0 * ((mean_len + 15) // 16)
        alg = {
            # a block of 256 representatives re-reads their packed words once per profile (from L2 after the first); 2 B per pair out
            "k_msv": U * wbytes + 2 * nprof * U,
            # the lazy stage's score-only Forward: per pair the packed read + the pair record in, 4 B out; nothing per row
            "k_fwd_bound": st["n_past_msv"] * (wbytes + 16 + 4),
            "k_bwd_bound": st.get("bwd_rows", 0) / max(mean_len, 1.0) * (wbytes + 16),
            # per pair: packed read + PairRec/PairOut; per row: 6 special-state floats written
            "k_filters_fwd": st["n_past_msv"] * (wbytes + 16 + 40) + st["fwd_rows"] * 24,
            # per row: Forward's 6 floats read, 6 decoding terms written
            "k_bwd_decode": st["n_past_fwd"] * (wbytes + 16 + 40) + st["fwd_rows"] * 48,
            # per row: 5 terms read (nothing written back)
            "k_decode": st["n_past_fwd"] * (16 + 40) + st["fwd_rows"] * 20,
            # envelope sweeps: Backward rows written once, read once (26 float4 per row), + 88 B result per envelope
            "k_env_fwd+k_env_bwd+k_env_post": st["env_rows"] * 2 * 26 * 16 + st["n_domains"] * (16 + 88),
        }
        # what bounds each of them: the DP scans are VALU-bound (no-FMA fp32), the streaming kernels HBM-bound
        rows = st["fwd_rows"]
        # lane-rows each scan kernel COMPUTED (with prefix sharing: the chains' own rows, not the rows of their pairs)
        krows = {"k_filters_fwd": rows, "k_bwd_decode": rows, "k_fwd_bound": st["bound_rows"], "k_msv": st["msv_rows"], "k_bwd_bound": st.get("bwd_rows", 0)}
        FLOPS_PER_ROW["k_fwd_bound"] = 16.0 * mean_nodes               # the profiles' own node count (45 for every ITSx model but two: 720), not the kernel's 46 slots
        FLOPS_PER_ROW["k_bwd_bound"] = 16.0 * mean_nodes               # the same recurrences transposed: the same count
        # the roofline and the issue fractions use a kernel's ALONE time when the extra step measured one (the timed steps' can be stretched by what ran beside it)
        kt = {k: (alone[k] if (alone and alone.get(k, 0) > 0) else kern[k]) for k in kern}
        tfl = {k: (krows[k] * FLOPS_PER_ROW[k] / (kt[k] * 1e-3) / 1e12 if kt[k] > 0 else None) for k in ("k_filters_fwd", "k_bwd_decode", "k_fwd_bound", "k_bwd_bound")}
        vfrac = {k: issue_frac(k, krows[k] / 64.0, kt[k]) for k in ISSUE_MIX}
        pmc = PMC_BYTES_PER_ROW or {}
        kernel_table = {k: {"ms": round(kern[k], 3), "ms_alone": round(alone[k], 3) if alone else None,
                            "alg_GBps": round(alg[k] / (kern[k] * 1e-3) / 1e9, 1) if kern[k] > 0 else None,
                            "hbm_frac": round(alg[k] / (kern[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kern[k] > 0 else None,
                            "nofma_tflops": round(tfl[k], 2) if tfl.get(k) else None,
                            "valu_issue_frac": round(vfrac[k], 3) if vfrac.get(k) else None,
                            # HBM traffic from the PMC counters (bytes per lane-row x the rows of this run) over the kernel's algorithmic bytes
                            "traffic_vs_alg": round(pmc[k] * krows[k] / alg[k], 2) if (k in pmc and k in krows and alg[k] > 0) else None} for k in kern}
        # launches of the dominant kernel in one step, for per-launch figures
        nl = {"k_msv": max(1, int(st.get("msv_launches", 1))), "k_fwd_bound": max(1, int(st.get("n_bound_launches", 1)))}.get(dom, max(1, int(st.get("n_batches", 1))))
        traffic = None
        if PMC_BYTES_PER_ROW and dom in PMC_BYTES_PER_ROW:
            traffic = PMC_BYTES_PER_ROW[dom] * krows.get(dom, rows) / nl
        # SURVEY 8d's algorithmic bytes of the WHOLE path per step: per read ceil(L / 4) + 48, per unique the packed words once per profile tile
        survey_bytes = n_local * (-(-mean_len // 4) + 48.0) + U * wbytes
        if dom in tfl and tfl[dom]:
            peak = VALU_FMA_TFLOPS if dom in USES_FMA else VALU_NOFMA_TFLOPS
            roof = {"kernel": dom, "bound": "valu", "achieved": round(tfl[dom], 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": tfl[dom] / peak, "frac_of_fma_peak": tfl[dom] / VALU_FMA_TFLOPS, "frac_of_nofma_peak": tfl[dom] / VALU_NOFMA_TFLOPS,
                    "fma_peak": VALU_FMA_TFLOPS, "nofma_peak": round(VALU_NOFMA_TFLOPS, 1), "valu_issue_frac": vfrac.get(dom),
                    "issue_ns": dict(ISSUE_NS), "issue_mix_per_row": ISSUE_MIX.get(dom), "issue_source": ISSUE_SOURCE,
                    "time_used": "alone (ITSX_MSV_OVERLAP=0 step)" if alone else "timed steps",
                    # what the same launches deliver counted on the rows of the kernel's PAIRS (the rows the chains skip included): not the
                    # kernel's own rate -- `achieved` is that -- but the rate an unshared kernel would need to match it
                    "equivalent_tflops_on_pairs_rows": (round(st["bound_rows_full"] * FLOPS_PER_ROW[dom] / (kt[dom] * 1e-3) / 1e12, 2)
                                                        if (dom == "k_fwd_bound" and st.get("bound_rows_full")) else None),
                    "launches_per_step": nl, "avg_launch_ms": kt[dom] / nl,
                    "alg_flops_per_launch": krows[dom] * FLOPS_PER_ROW[dom] / nl, "alg_flops_per_lane_row": FLOPS_PER_ROW[dom], "alg_bytes_per_launch": alg[dom] / nl, "traffic": traffic,
                    # what the kernel EXECUTES: 8.5 of the 11 operations per node (three folded into the profile's table), every one counted as its
                    # flops (a fused multiply-add two): the fraction of the peak that the instructions actually issued account for
                    "executed_flops_per_lane_row": round(FLOPS_PER_ROW[dom] * 8.5 / 11.0, 1) if dom == "k_fwd_bound" else None,
                    "frac_executed": (tfl[dom] / peak) * 8.5 / 11.0 if dom == "k_fwd_bound" else None,
                    # SURVEY 8d's figure for the WHOLE path: its algorithmic bytes per step over the step's wall time, against the HBM peak -- the path is
                    # not HBM-bound (its DP stages are scans bound by VALU issue); reported because 8d asks for it
                    "hbm_frac_whole_path_on_survey_bytes": survey_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                    "hbm_frac_on_alg_bytes": alg[dom] / (kern[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic_source": PMC_SOURCE if traffic is not None else None,
                    "survey_bytes_per_step": survey_bytes,
                    "traffic_vs_survey_bytes": (traffic * nl / survey_bytes) if traffic is not None else None,
                    "note": ("the lazy stage's score-only Forward over the rows its chains COMPUTE (prefix sharing: config.rows_shared_frac of the pairs' rows come "
                             "from saved states): the recurrence's own flops per lane-row in node order (16 per model node: 720 for the 45-node ITSx models) "
                             "against the fp32 FMA peak 157.3 TFLOP/s (ALGORITHMIC flops: the kernel itself executes 8.5 of the 11 operations per node -- the deletes' scale and their share of E are folded into the profile's table, DESIGN.md 5e) -- 139 of its 262 instructions per row are fused, and a packed instruction issues at the "
                             "rate of two plain ones (profiles/round5_valu_issue.md), so valu_issue_frac -- the kernel's instruction mix priced with the measured "
                             "issue times, over the SIMDs' time -- is the figure that says how much is left; duration = HIP events on the engine's stream.  " if dom in USES_FMA else "") +
                            "a serial recurrence per (representative, profile): the recurrence's own no-FMA fp32 flops per lane-row (HMMER rounds "
                            "products and sums separately; Forward 960, Backward 1056 with its four unconditional DD passes: the count is "
                            "spelled out at the top of bench.py) against 78.6 TFLOP/s = 1024 SIMDs x 32 lanes x 2.4 GHz; duration = HIP events on the "
                            "engine's stream; traffic = PMC bytes per lane-row x rows per launch (traffic_source); traffic_vs_survey_bytes = the kernel's HBM traffic per step "
                            "over SURVEY 8d's algorithmic bytes of the whole path.  "
                            "hbm_frac_on_alg_bytes / alg_bytes_per_launch count the DP slab rows (24 B written + 24 B read per lane-row) as the kernel's "
                            "algorithmic bytes: a cost of this design (rows handed from Forward to Backward to the decoder through HBM), ~14 000 x "
                            "SURVEY 8d's per-read bytes for the whole path -- beside the kernel's VALU bound, not a measure of efficiency"}
        else:
            ach = alg[dom] / (kern[dom] * 1e-3) / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "launches_per_step": nl, "avg_launch_ms": kern[dom] / nl, "alg_bytes_per_launch": alg[dom] / nl, "traffic": traffic}
        c_start, c_stop = (out[0][0], out[0][1]) if isinstance(out[0], tuple) else (out[0][:, 0], out[0][:, 1])
        trimmed = int(((c_start >= 0) & (c_stop >= 0) & (c_start < c_stop)).sum())
        shape = ("merged reads of 300-580 bases (mean %.0f)" % mean_len) if cfg2 else \
            ("2x250-merged reads of 300-480 bases (mean %.0f)" % mean_len) if cfg4 else "300 bp single-end reads"
        cname = "configs[3] (one GPU's share of the 50 M reads)" if cfg3 else "configs[2]" if cfg2 else ("configs[4] (one GPU's shard; the full share is 12.5 M reads)" if cfg4 else "configs[1]")
        wl = cname + ": %d synthetic %s per GPU" % (n_local, shape)
        if strong:
            wl = cname + ": %d synthetic %s in total, sharded over %d ranks (shared template library)" % (args.total_reads, shape, world)
        wl += (", --region ALL (1_ / 4_ profiles)" if cfg4 else ", ITS2") + \
            (", cluster_id=1.0 (pure derep)" if args.cluster_id >= 1.0 else ", cluster_id=%g (greedy clustering, row a2)" % args.cluster_id)
        if args.taxa == "all":
            wl += ", --taxa All (814 profiles, as configs[3])"
        res = {
            "metric": "reads/sec trimmed (%s%s)" % ("Fungi " if fungi else "", ("region ALL" if cfg4 else "ITS2") + ("" if fungi else ", stand-in taxon Tracheophyta for Fungi")),
            "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": warm_done, "warmup_requested": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8+f32", "data": "synthetic",
            "config": {"workload": wl,
                       "taxon": ("Fungi (%s)" % fungi) if fungi else "Tracheophyta (stand-in: F.hmm absent from the reference mount)" if args.taxa == "T" else "All (every ITSx set in the mount; F.hmm absent)", "profiles": nprof,
                       "reads_rank0": n_local, "mean_length": round(mean_len, 1), "generate_s": round(t_gen, 1),
                       "unique": int(st["n_unique"]), "pairs_past_msv": int(st["n_past_msv"]), "pairs_past_fwd": int(st["n_past_fwd"]),
                       "domains": int(st["n_domains"]), "domain_rows_resident": int(st["n_rows_resident"]),
                       "domain_rows_resident_GB": round(st["n_rows_resident"] * 80 / 1e9, 2), "reads_trimmed_rank0": trimmed,
                       "domain_stage": {"lazy": "lazy (exact): pairs that cannot win ItsPosition's argmax stop after their Forward score", "compact": "every pair evaluated, rows compacted",
                                        "full": "every pair evaluated, every row resident"}[args.rows] if not st["n_lazy_reruns"] else "lazy -> repeated in full (rows depended on the exact domZ)",
                       "pairs_evaluated": int(st["n_lazy_evaluated"]) if st["lazy"] else int(st["n_past_msv"]),
                       "pairs_evaluated_round1": int(st["n_lazy_round1"]) if st["lazy"] else None,
                       "undecided_rows_that_mattered": int(st["n_lazy_pending"]) if st["lazy"] else None,
                       "profiles_counted_exactly": int(st["n_lazy_completed_profiles"]) if st["lazy"] else None,
                       "pairs_of_those_profiles": int(st["n_lazy_completed"]) if st["lazy"] else None,
                       # prefix sharing (csrc/k_share.hip): rows the two scan kernels did NOT compute because another representative of the same length
                       # had the same first residues (the state comes from its saved row state); tree = what the prefix tree offers at this block size
                       # pass A's rows (Forward chains + Backward chains) over the rows of its pairs: what two-sided sharing leaves to compute
                       "rows_computed_frac": round((st["bound_rows"] + st.get("bwd_rows", 0)) / st["bound_rows_full"], 4) if st.get("bound_rows_full") else None,
                       "lane_rows": {k: int(st[k]) for k in ("bound_rows", "bwd_rows", "bound_rows_full", "msv_rows", "msv_rows_full")},
                       "two_sided": ({"joined_representatives": int(st["n_joined"]), "backward_chains": int(st["bwd_chains"]), "backward_states": int(st["gamma_nodes"]),
                                      "forward_rows_frac": round(st["bound_rows"] / st["bound_rows_full"], 4) if st.get("bound_rows_full") else None,
                                      "backward_rows_frac": round(st["bwd_rows"] / st["bound_rows_full"], 4) if st.get("bound_rows_full") else None,
                                      "backward_launches": int(st["n_bwd_launches"])} if st.get("two_sided") else None),
                       "rows_shared_frac": {"k_fwd_bound": round(1.0 - st["bound_rows"] / st["bound_rows_full"], 4) if st.get("bound_rows_full") else None,
                                            "k_msv": round(1.0 - st["msv_rows"] / st["msv_rows_full"], 4) if st.get("msv_rows_full") else None,
                                            "tree": round(st["share_frac"], 4), "block_rows": int(st["share_B"]), "saved_states": int(st["share_nodes"]),
                                            "chains_with_a_parent": int(st["share_chains"]), "batches": int(st["share_batches"]),
                                            "pairs_run_for_their_states_only": int(st["n_share_helpers"]), "build_ms": round(acc.get("ms_share_build", 0.0) / K, 2)},
                       # the library's environment switches that were set when the last timed search started (csrc/switches.cpp; only non-default ones
                       # can appear: an empty dict = every switch at its default) and the Python layer's
                       "switches": {**eng.switches(), **{k: v for k, v in os.environ.items() if k.startswith("ITSXPRESS_")}},
                       "parallelism": "reads sharded x%d%s" % (world, (", exact global derep" if args.global_derep else ", per-shard derep") if use_dist else "")},
            "full_pipeline": full_leg,
            "full_pipeline_value": (total_local * full_leg["steps"] / (full_leg["ms_per_step"] * 1e-3 * full_leg["steps"])) if (full_leg and "ms_per_step" in full_leg) else None,
            "timed_region": "ASCII text resident in HBM -> device 2-bit packing -> derep -> MSV -> Forward/Backward -> domains -> "
                            "thresholds -> per-read coordinates on the host (+ all-reduce / gather at N > 1)",
            "host_handover": handover,
            "winners_files": files_leg,
            "paired_file_to_file": paired_leg,
            "ranks": rank_ms,
            "stage_ms": {k: round(v / K, 3) for k, v in acc.items()},
            # how stage_ms adds up to a step (its entries are NOT disjoint): the top-level stages are ms_pack + ms_derep + ms_msv + ms_filters + ms_domains +
            # ms_finalize; ms_lazy_complete is part of ms_finalize AND its own filter / domain work is accumulated into ms_msv / ms_filters / ms_domains a second
            # time, so it is subtracted once; every *_kernel / ms_ensemble / ms_lazy_select / ms_share_build entry lies inside one of the stages (ms_bound_kernel,
            # ms_bwd_bound, ms_fwd_kernel, ms_lazy_select in ms_filters; ms_bwd / decode / env / ensemble in ms_domains); ms_msv_kernel of chunk c + 1 and
            # ms_bias_kernel run beside other kernels (stretched wall times); the rest of the step is host work (result copies, the coordinates' hand-over)
            "stage_overlap": {"top_level_ms": round((sum(acc.get(k, 0.0) for k in ("ms_pack", "ms_derep", "ms_msv", "ms_filters", "ms_domains", "ms_finalize", "ms_cluster", "ms_merge"))
                                                     - acc.get("ms_lazy_complete", 0.0) - acc.get("ms_lazy_topup_stages", 0.0)) / K, 1),
                              "ms_per_step": round(dt / args.steps * 1e3, 1),
                              "host_calls_ms": host_calls_ms,
                              "note": "host_calls_ms = wall time of each engine call of a step as the host sees it (they add up to ms_per_step: what the stages' own "
                                      "timers leave out is inside set_reads / trim_coords -- offsets in, coordinates out); top_level_ms = ms_pack + ms_derep + ms_msv + ms_filters + ms_domains + ms_finalize - ms_lazy_complete - ms_lazy_topup_stages (both "
                                      "counted in ms_finalize and, through the stages they re-run, in ms_msv / ms_filters / ms_domains); all other stage_ms entries are nested in these; the "
                                      "difference to ms_per_step is host time between the stages"},
            "kernels": kernel_table,
            "parity_risk": {"regions": int(st["n_regions"]), "regions_multidomain": int(st["n_multidomain"]),
                            "uniques_winner_is_cluster_envelope": int(st["n_uniq_multi_winner"]),
                            "reads_winner_is_cluster_envelope": int(st["n_reads_multi_winner"]),
                            "regions_clustered": int(st["n_mr_clustered"]), "distinct_regions_sampled": int(st["n_mr_distinct"]), "regions_clustering_failed": int(st["n_mr_failed"]),
                            "cluster_envelopes": int(st["n_mr_envelopes"]),
                            "pairs_over_region_cap": int(st["n_domain_overflow"]),
                            "uniques_with_pair_over_region_cap": int(st["n_uniq_region_cap"]),
                            "reads_with_pair_over_region_cap": int(st["n_reads_region_cap"])},
            "concurrency": "k_bias of batch b+1 runs on a second stream beside k_decode of batch b, and k_msv of chunk c+1 beside the domain stage of chunk c: ms_bias_kernel and ms_msv_kernel (hence kernels.k_msv and valu.msv_gcups) are stretched wall times, not extra step time; alone k_msv takes ~0.2 s per chunk (ITSX_MSV_OVERLAP=0)",
            "cluster": None if args.cluster_id >= 1.0 else {"windows": int(st["cl_windows"]), "cut_windows": int(st["cl_cuts"]),
                                                            "alignments": int(st["cl_alignments"]), "certified_rejections": int(st["cl_certified"]),
                                                            "centroids": int(st["n_unique"]), "ms_per_step": round(acc.get("ms_cluster", 0.0) / K, 1)},
            # stage A ("the genuinely HBM-bound stage"): device packing of the resident text + dereplication (or clustering),
            # on SURVEY's algorithmic bytes per read, ceil(L / 4) + 48 (+ the L bytes of ASCII text the packing kernel reads here)
            "stage_a": stage_a(acc, K, n_local, mean_len, args.cluster_id),
            "roofline": roof,
            "valu": {"msv_gcups": st["msv_cells"] / (kern["k_msv"] * 1e-3) / 1e9 if kern["k_msv"] > 0 else None,
                     "fwd_rows_per_s": rows / (kern["k_filters_fwd"] * 1e-3) if kern["k_filters_fwd"] > 0 else None,
                     "bwd_rows_per_s": rows / (kern["k_bwd_decode"] * 1e-3) if kern["k_bwd_decode"] > 0 else None,
                     "env_rows_per_s_x3_sweeps": 3 * st["env_rows"] / (kern["k_env_fwd+k_env_bwd+k_env_post"] * 1e-3) if kern["k_env_fwd+k_env_bwd+k_env_post"] > 0 else None,
                     "peak_nofma_tflops": VALU_NOFMA_TFLOPS},
        }
        if args.cpu_sample != 0 and world == 1:          # the CPU baseline is a rank-0, N=1 leg only
            threads, cpu_limit = effective_cpus()        # usable CPUs, not the host's hardware threads
            if args.cpu_sample < 0:
                # ~20 s of CPU work: the SSE2 port does ~70 (440-base reads x 155 profiles) .. 130 (300-base) reads/s per thread
                rate = (130.0 if args.workload == "cfg1" else 70.0) * (155.0 / max(nprof, 1))
                args.cpu_sample = int(min(40000, max(600, 20.0 * rate * threads)))
                room = args.budget_s - (time.time() - T_START) - 20.0     # less when the budget is nearly spent
                args.cpu_sample = int(max(600, min(args.cpu_sample, args.cpu_sample * max(room, 0.0) / 30.0)))
                if args.cluster_id < 1.0:                # the baseline's greedy clustering is sequential by definition, and quadratic: 3 000 reads take ~15-45 s
                    args.cpu_sample = min(args.cpu_sample, 3000)
            m = min(args.cpu_sample, n_local)
            progress("CPU baseline on %d reads, %d threads" % (m, threads))
            v, cdt, nc, ccoords, seqs, cextra = cpu_baseline(hmm, blob, offs, m, threads, lp, rp, args.cluster_id)
            progress("CPU baseline done in %.1f s" % cdt)
            # trim-coordinate concordance (BASELINE metric): the engine on the very same sample against the baseline path
            e2 = Engine(local_rank)
            e2.set_rows_mode(args.rows)                   # the benchmarked configuration is the one compared
            e2.load_profiles(text=hmm)
            e2.set_reads_buffer(np.ascontiguousarray(blob[:int(offs[m])]), offs[:m + 1])
            if args.cluster_id < 1.0:
                e2.cluster(args.cluster_id, strand_both=True)
            else:
                e2.derep(strand_both=True, minseqlength=1)
            e2.search(T=10.0, F1=1e-6, F2=1e-6, F3=1e-6)
            e2.finalize(domE=10.0)
            gs, ge, gt, _ = e2.trim_coords(lp, rp)
            e2.close()
            got = np.stack([gs, ge, gt], axis=1)
            conc = float((got == ccoords).all(axis=1).mean())
            res["cpu_baseline"] = {"value": v, "unit": "reads/s", "cores": threads, "kind": "port",
                                   "sample": "first %d reads of the same workload (%d %s), %s+search+argmax, %.1f s" %
                                             (m, nc, "centroids" if args.cluster_id < 1.0 else "unique", "cluster_size" if args.cluster_id < 1.0 else "derep", cdt),
                                   "trim_coord_concordance": conc,
                                   "concordance_is": "engine vs oracle/ (our restatement), not vs vsearch+hmmsearch"}
            res["cpu_baseline"].update(cextra)
            res["cpu_baseline"].update({"host_hardware_threads": os.cpu_count(), "cores_limited_by": cpu_limit})
            # the real reference engines, when the box happens to have them (SURVEY 8d "preferred")
            import tempfile
            with tempfile.TemporaryDirectory() as tmp:
                try:
                    real = real_tools_baseline(hmm, seqs, threads, tmp) if (args.cluster_id >= 1.0 and not cfg4) else None
                except Exception as e:      # a tool that is present but fails is reported, not fatal
                    real = None
                    res["cpu_baseline"]["real_tools_error"] = repr(e)[:200]
            if real is not None:
                rdt, rcoords = real
                res["cpu_baseline"].update({"value": m / rdt, "kind": "reference", "cores": threads,
                                            "sample": "first %d reads of the same workload through vsearch --fastx_uniques + hmmsearch (--cpu %d) + "
                                                      "the reference's parsers, %.1f s" % (m, threads, rdt),
                                            "trim_coord_concordance": float((got == rcoords).all(axis=1).mean()),
                                            "concordance_is": "engine vs the real vsearch + hmmsearch path",
                                            "port_value": v, "port_concordance": conc})
        # N > 1: is the sharded job's answer the one-GPU answer?  Rank 0 regenerates every rank's shard, runs ONE engine on the whole
        # job and compares every coordinate row for equality (north_star: "100 % trim-coordinate concordance").
        if use_dist and out is not None and args.cluster_id >= 1.0:
            tot = int(total_local)
            room = args.budget_s - (time.time() - T_START)
            if not args.concordance_reads or tot > args.concordance_reads:
                res["concordance_vs_single_engine"] = {"skipped": "the job has %d reads, --concordance-reads is %d" % (tot, args.concordance_reads)}
            elif room < 40.0 + 6.0 * dt / max(args.steps, 1) * world:
                res["concordance_vs_single_engine"] = {"skipped": "wall-clock budget (--budget-s %.0f): %.0f s left" % (args.budget_s, room)}
            else:
                tcz = time.time()
                progress("one engine on the whole job (%d reads) for concordance_vs_single_engine" % tot)
                parts, lens = [], []
                for r in range(world):
                    n_r = shard_plan(args.workload, args.reads, args.total_reads, args.weak, world, r)[0]
                    gr = dict(gen, seed=synth.SEED + cfgno + 1000 * r)
                    if strong:
                        gr.update(frac_templates=0.02 * args.total_reads / max(n_r, 1))
                    b_r, o_r = synth.make_reads(thmm, n_r, **gr)
                    parts.append(b_r)
                    lens.append(np.diff(o_r))
                wblob = np.concatenate(parts)
                woffs = np.zeros(tot + 1, np.int64)
                np.cumsum(np.concatenate(lens), out=woffs[1:])
                del parts
                e1 = Engine(local_rank)
                e1.set_rows_mode(args.rows)
                e1.load_profiles(text=hmm)
                e1.set_reads_buffer(wblob, woffs)
                e1.derep(strand_both=True, minseqlength=1)
                e1.search(T=10.0, F1=1e-6, F2=1e-6, F3=1e-6)
                e1.finalize(domE=10.0)
                one = np.stack(e1.trim_coords(lp, rp), axis=1)
                e1.close()
                got = np.concatenate(out)
                same = one.shape == got.shape and bool(np.array_equal(one, got))
                res["concordance_vs_single_engine"] = {
                    "reads": tot, "equal": same,
                    "fraction_of_reads_equal": float((one == got).all(axis=1).mean()) if one.shape == got.shape else 0.0,
                    "derep": "exact global (hash-partitioned all-to-all)" if args.global_derep else "per shard (--per-shard-derep)",
                    "seconds": round(time.time() - tcz, 1),
                    "note": "rank 0 regenerated every rank's shard and ran one engine on the whole job: start, stop, tlen and the 'has a row' flag of every read"}
        os.write(REAL_STDOUT, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
