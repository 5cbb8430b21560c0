#!/bin/bash
# k_bwd_bound with parts switched off (diagnostic bits: results are garbage and the search may fail after pass A -- the kernel trace is what counts)
export TMPDIR=/tmp
out=$PWD/gpurun_out/exp5; mkdir -p $out
common="--steps 1 --warmup 0 --cpu-sample 0 --handover-steps 0 --full-steps 0 --alone-steps 0"
for arm in ${ARMS:-0 8 16 24 32 40}; do
  rm -rf $out/trace
  ITSX_TEST_HOOKS=1 ITSX_PASSA_DBG=$arm rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py $common > $out/line_$arm.json 2> $out/err_$arm.log
  f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$arm" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def tot(name):
    r = [x for x in rows if name in x["Kernel_Name"]]
    return len(r), sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in r) / 1e6
print("dbg", sys.argv[2], "k_bwd_bound n %d %.1f ms" % tot("k_bwd_bound"), "| k_fwd_bound n %d %.1f ms" % tot("k_fwd_bound"), "| k_msv_bwd n %d %.1f ms" % tot("k_msv_bwd"), "| k_msv< n %d %.1f ms" % tot("k_msv<"), flush=True)
PY
done
rm -rf $out/trace
