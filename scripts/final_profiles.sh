#!/bin/bash
# the round's final measurements, one box: bash scripts/final_profiles.sh  (writes gpurun_out/final/)
set -o pipefail
export TMPDIR=/tmp
out=$PWD/gpurun_out/final; mkdir -p $out
lean="--cpu-sample 0 --handover-steps 0 --full-steps 0 --files-leg 0"
echo "[final] default bench"; python bench.py > $out/bench_default_line.json 2> $out/bench_default.err || { echo FAILED default; tail -5 $out/bench_default.err; }
echo "[final] kernel stats"; rm -rf $out/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 1 --warmup 1 $lean --alone-steps 0 > $out/bench_10M_profiled_line.json 2> $out/stats.err
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/bench_10M_kernel_stats.csv; rm -rf $out/stats
echo "[final] cfg1"; python bench.py --workload cfg1 $lean > $out/bench_cfg1_line.json 2> $out/cfg1.err
echo "[final] 1250k"; python bench.py --reads 1250000 --steps 5 --warmup 2 $lean > $out/bench_1250k_line.json 2> $out/1250k.err
echo "[final] cfg3"; python bench.py --workload cfg3 --steps 2 --warmup 1 $lean > $out/bench_cfg3_line.json 2> $out/cfg3.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final/*_line.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(f.split("/")[-1], round(d["value"]), d["unit"], round(d["ms_per_step"], 1), "ms", "roofline", d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("avg_launch_ms"))
    except Exception as e:
        print(f, "unreadable", e)
PY
