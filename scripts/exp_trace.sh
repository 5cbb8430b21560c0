#!/bin/bash
# kernel trace of one bench step (every dispatch with its grid and duration) + pass A's launch dump
set -o pipefail
out=$PWD/gpurun_out/exp2; mkdir -p $out
export TMPDIR=/tmp
common="--steps 1 --warmup 1 --cpu-sample 0 --handover-steps 0 --full-steps 0 --alone-steps 0"
${DUMP:+env ITSX_PASSA_DUMP=1 }rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py $common > $out/line.json 2> $out/err.log
ls -R $out/trace | head -20
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$out" <<'PY'
import csv, sys, collections
f, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
print(len(rows), "dispatches; columns:", list(rows[0].keys()))
# keep a compact version: name (short), start, end, grid, workgroup
with open(out + "/dispatches.tsv", "w") as o:
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void itsx::", "").replace("itsx::", "")
        o.write("\t".join([name, r["Start_Timestamp"], r["End_Timestamp"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("Stream_Id", r.get("Queue_Id", ""))]) + "\n")
PY
grep "^\[passa\]" $out/err.log > $out/passa_dump.txt; wc -l $out/passa_dump.txt
rm -rf $out/trace
