#!/bin/bash
# gpurun -- 'bash tools/r2_wholetree.sh [variant ...]': whole-tree sweep leg (crowns and walk off) per build variant
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_wholetree
mkdir -p "$OUT"
cd "$REPO"
ARGS="--no-crowns --no-walk --steps 3 --warmup 1 --no-cpu-baseline --no-sensitivity --pcie-steps 0 --roofline-steps 0"
for v in "$@"; do
  lib=variants/$v/libwepp_place.so; [ "$v" = head ] && lib=wepp_amd/libwepp_place.so
  WEPP_PLACE_LIB=$lib timeout -k 10 400 python bench.py $ARGS > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err" || { echo "$v failed"; tail -3 "$OUT/bench_$v.err"; exit 1; }
  python - <<PY
import json
b=json.load(open("$OUT/bench_$v.json"))
print("$v: %.4g reads/s  %.1f ms/step  alg %.0f GB/s" % (b["value"], b["ms_per_step"], b["roofline_hbm"]["achieved"]))
PY
done
