#!/bin/bash
# build_variant.sh NAME "EXTRA FLAGS"  ->  build/NAME/libitsx_hip.so : a second build of the engine with extra compiler flags
# (for same-box A/B runs with scripts/ab_bench.py; the in-tree library is untouched)
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/itsxpress_amd/csrc
OUT=$ROOT/build/$NAME
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fgpu-flush-denormals-to-zero -Wall -Wno-unused-function -Wno-unused-result $EXTRA"
pids=""
for f in engine k_util k_derep k_msv k_vit k_float k_ensemble k_cluster k_merge; do
  /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $OUT/$f.o & pids="$pids $!"
done
for f in hmm_host trim_host fastq_io pinflate; do
  /opt/rocm/bin/hipcc $FLAGS -x hip -c $SRC/$f.cpp -o $OUT/$f.o & pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libitsx_hip.so $OUT/*.o -lz -lpthread -ldl
echo $OUT/libitsx_hip.so
