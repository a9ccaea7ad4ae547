#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: builds variants/<name>/libwepp_place.so from the same
# sources with extra compile flags (profiling / experiment builds; load with WEPP_PLACE_LIB=...).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/wepp_amd/csrc
OUT=$ROOT/variants/$NAME
mkdir -p "$OUT"
FLAGS="-O3 -std=c++17 -fPIC -Wno-unused-parameter $*"
pids=()
for f in flatmat gen errors flat_debug capi fitch_capi epp_capi; do
  /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.cpp -o $OUT/$f.o & pids+=($!)
done
for f in place_kernels sort_reads fitch_kernels epp_kernels; do
  /opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 -c $SRC/$f.hip -o $OUT/$f.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libwepp_place.so $OUT/*.o
rm -f $OUT/*.o
echo "built $OUT/libwepp_place.so"
