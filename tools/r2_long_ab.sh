#!/bin/bash
# gpurun -- 'bash tools/r2_long_ab.sh': long-read bench with and without per-event bounds on the window streams
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_long_ab
mkdir -p "$OUT"
cd "$REPO"
ARGS="--read-len 1200 --reads 200000 --steps 3 --warmup 1 --no-cpu-baseline --no-sensitivity --pcie-steps 0 --roofline-steps 0"
for e in 0 1; do
  WEPP_WIN_EAGER=$e timeout -k 10 300 python bench.py $ARGS > "$OUT/bench_e$e.json" 2> "$OUT/bench_e$e.err" || exit 1
  python - <<PY
import json
b=json.load(open("$OUT/bench_e$e.json"))
print("eager=$e: %.4g reads/s  %.1f ms/step" % (b["value"], b["ms_per_step"]))
PY
done
WEPP_PLACE_LIB=variants/stats/libwepp_place.so timeout -k 10 300 python tools/sweep_stats.py 16000000 200000 1200 > "$OUT/stats_e1.txt" 2> "$OUT/stats.err"
echo "stats rc=$?"
