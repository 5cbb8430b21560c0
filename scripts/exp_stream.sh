#!/bin/bash
# arms of the streamed single-end file run over ONE generated file: bash scripts/exp_stream.sh name "ENV=.." ...
set -o pipefail
out=gpurun_out/exp_stream; mkdir -p $out
dir=/tmp/itsx_inputs
while [ $# -ge 2 ]; do
  name=$1; envs=$2; shift 2
  env $envs python scripts/file_run.py --reads ${READS:-10000000} --shape cfg2 --stream --stream-write --input-dir $dir > $out/$name.json 2> $out/$name.err || { echo "FAILED $name"; tail -5 $out/$name.err; exit 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
tl = d["stream_timeline_s(chunk, text ready, loaded, searched)"]
print(sys.argv[2], d["s_total"], d["stages_s"], "counting", d["finalize_s"].get("counting"), "chunks", d["stream_chunks"], "last text/searched", tl[-1][1], tl[-1][3], flush=True)
PY
done
