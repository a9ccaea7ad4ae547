#!/bin/bash
# tools/build_variant.sh <name> [--patch FILE]... [extra hipcc flags]: builds variants/<name>/libwepp_place.so from
# the product sources with extra compile flags and, optionally, patches applied to a scratch copy of the sources
# (tools/variants/*.patch: experiment code that does not belong in the product source, e.g. the timing-only builds
# with WRONG results).  Load the result with WEPP_PLACE_LIB=variants/<name>/libwepp_place.so.
#   tools/build_variant.sh stats -DWEPP_SWEEP_STATS
#   tools/build_variant.sh nohits --patch tools/variants/timing_experiments.patch -DWEPP_EXP_NO_HITS
#   tools/build_variant.sh win64 -DWEPP_WIN_SIZE=64 -DWEPP_WIN_STRIDE=32      (tests/test_gpu_parity.py: window fuzz)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/variants/$NAME
mkdir -p "$OUT"
SCRATCH=$(mktemp -d)
trap 'rm -rf "$SCRATCH"' EXIT
mkdir -p "$SCRATCH/wepp_amd" "$SCRATCH/include"
cp -r "$ROOT/wepp_amd/csrc" "$SCRATCH/wepp_amd/csrc"
cp "$ROOT/include/wepp_place.h" "$SCRATCH/include/"
rm -f "$SCRATCH"/wepp_amd/csrc/*.o
EXTRA=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--patch" ]; then (cd "$SCRATCH" && patch -p0 -s < "$(cd "$ROOT" && realpath "$2")"); shift 2
  else EXTRA+=("$1"); shift; fi
done
SRC=$SCRATCH/wepp_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC -Wno-unused-parameter ${EXTRA[*]}"
pids=()
for f in flatmat gen errors flat_debug flat_io capi fitch_capi epp_capi; do
  /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.cpp -o $OUT/$f.o & pids+=($!)
done
for f in $(cd $SRC && ls *.hip | sed 's/\.hip$//'); do
  /opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 -c $SRC/$f.hip -o $OUT/$f.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libwepp_place.so $OUT/*.o
rm -f $OUT/*.o
echo "built $OUT/libwepp_place.so"
