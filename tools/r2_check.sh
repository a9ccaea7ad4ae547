#!/bin/bash
# gpurun -- 'bash tools/r2_check.sh': the whole GPU test suite, then the default bench line
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_check
mkdir -p "$OUT"
cd "$REPO"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > "$OUT/pytest_gpu.log" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 "$OUT/pytest_gpu.log"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench rc=$?"; tail -3 "$OUT/bench_default.err"; cut -c1-600 "$OUT/bench_default.json"
