#!/bin/bash
# gpurun -- 'bash tools/r2_long_variants.sh v1 v2 ...': long-read bench per build variant (variants/<v>/; "head" = the product build)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_long_variants
mkdir -p "$OUT"; cd "$REPO"
ARGS="--read-len 1200 --reads 200000 --steps 3 --warmup 1 --no-cpu-baseline --no-sensitivity --pcie-steps 0 --roofline-steps 0"
for v in "$@"; do
  lib=variants/$v/libwepp_place.so; [ "$v" = head ] && lib=wepp_amd/libwepp_place.so
  WEPP_PLACE_LIB=$lib timeout -k 10 300 python bench.py $ARGS > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err" || { echo "$v failed"; tail -3 "$OUT/bench_$v.err"; exit 1; }
  python - <<PY
import json
b=json.load(open("$OUT/bench_$v.json"))
print("$v: %.4g reads/s  %.1f ms/step" % (b["value"], b["ms_per_step"]))
PY
done
