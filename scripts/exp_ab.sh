#!/bin/bash
# A/B arms of the default bench step: bash scripts/exp_ab.sh name1 "ENV=.. ENV=.." name2 "..." ...
set -o pipefail
out=gpurun_out/exp_ab; mkdir -p $out
common="${BENCH_ARGS:---steps 3 --warmup 1 --cpu-sample 0 --handover-steps 0 --full-steps 0 --files-leg 0 --alone-steps 0}"
while [ $# -ge 2 ]; do
  name=$1; envs=$2; shift 2
  env $envs python bench.py $common > $out/$name.json 2> $out/$name.err || { echo "FAILED $name"; tail -5 $out/$name.err; exit 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
s = d["stage_ms"]
keys = ["ms_msv", "ms_bound_kernel", "ms_bwd_bound", "ms_fwd_kernel", "ms_bwd_kernel", "ms_ensemble", "ms_domains", "ms_finalize", "ms_lazy_complete", "ms_lazy_topup", "ms_share_build"]
print(sys.argv[2], "ms_per_step", round(d["ms_per_step"], 1), " ".join("%s %.1f" % (k[3:], s[k]) for k in keys), "switches", d["config"].get("switches"), flush=True)
PY
done
