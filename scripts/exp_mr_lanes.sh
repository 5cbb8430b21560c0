#!/bin/bash
# ensemble stage: lanes per wave in lazy mode (few distinct regions per batch).  Usage: gpurun -- bash scripts/exp_mr_lanes.sh
set -o pipefail
out=gpurun_out/exp1; mkdir -p $out
common="--steps 2 --warmup 1 --cpu-sample 0 --handover-steps 0 --full-steps 0 --alone-steps 0"
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py $common > $out/$name.json 2> $out/$name.err || { echo "FAILED $name"; tail -5 $out/$name.err; return 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
s = d["stage_ms"]
print(sys.argv[2], "ms_per_step", round(d["ms_per_step"], 1), "ensemble", s["ms_ensemble"], "domains", s["ms_domains"], "finalize", s["ms_finalize"], "complete", s["ms_lazy_complete"], "topup", s["ms_lazy_topup"], flush=True)
PY
}
run base ITSX_NOP=1 &&
run lanes16 ITSX_MR_LONG_FRAC=1.0 ITSX_MR_LONG_LANES=16 &&
run lanes8 ITSX_MR_LONG_FRAC=1.0 ITSX_MR_LONG_LANES=8 &&
run lanes4 ITSX_MR_LONG_FRAC=1.0 ITSX_MR_LONG_LANES=4 &&
run lanes2 ITSX_MR_LONG_FRAC=1.0 ITSX_MR_LONG_LANES=2 &&
ITSX_MR_DEBUG=1 python bench.py $common > $out/dbg.json 2> $out/dbg.err; grep "ensemble batch" $out/dbg.err | head -20
