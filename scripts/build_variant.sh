#!/bin/bash
# build_variant.sh NAME "EXTRA FLAGS" [file ...]  ->  build/NAME/libitsx_hip.so : a second build of the engine in which the named sources
# (default: k_lazy) are compiled with extra flags (-D switches of the kernels) and every other object is the in-tree one.  For same-box
# A/B runs: ITSX_LIB=build/NAME/libitsx_hip.so python bench.py ...  (itsxpress_amd/_lib.py honours ITSX_LIB).
set -e
NAME=$1; EXTRA=$2; shift; shift
FILES=${@:-k_lazy}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/itsxpress_amd/csrc
OUT=$ROOT/build/$NAME
mkdir -p $OUT
make -s -j8 -C $SRC
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fgpu-flush-denormals-to-zero -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-variable $EXTRA"
OBJS=""
for o in $SRC/*.o; do
  b=$(basename $o .o); skip=0
  for f in $FILES; do [ "$f" = "$b" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
for f in $FILES; do
  if [ -f $SRC/$f.hip ]; then /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $OUT/$f.o; else /opt/rocm/bin/hipcc $FLAGS -x hip -c $SRC/$f.cpp -o $OUT/$f.o; fi
  OBJS="$OBJS $OUT/$f.o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libitsx_hip.so $OBJS -lz -lpthread -ldl
echo $OUT/libitsx_hip.so
