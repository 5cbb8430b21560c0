#!/bin/bash
# HBM traffic counters of the round's final kernels (each counter its own run, the program itself after `--`): gpurun_out/pmc6/
export TMPDIR=/tmp
R=$PWD
for tag in 1M 10M; do
  reads=$([ $tag = 1M ] && echo "--reads 1000000" || echo "")
  d=$R/gpurun_out/pmc6_$tag; rm -rf $d; mkdir -p $d
  for c in FETCH_SIZE WRITE_SIZE; do
    n=$([ $c = FETCH_SIZE ] && echo fetch || echo write)
    echo "[pmc] $tag $c"
    (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d/$n -- python3 $R/bench.py $reads --steps 1 --warmup 0 --cpu-sample 0 --handover-steps 0 --full-steps 0 --files-leg 0 --alone-steps 0 > $d/$n.json 2> $d/$n.err)
    find $d/$n -name "*kernel_trace.csv" -delete; find $d/$n -name "*agent_info.csv" -delete
  done
  du -sh $d
done
