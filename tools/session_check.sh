#!/bin/bash
# gpurun -- 'bash tools/session_check.sh': GPU parity tests, the default bench line, and a kernel trace
# of the same step with one sweep launch per stream (WEPP_SWEEP_UNFUSED=1) for a per-stream breakdown.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/check
mkdir -p "$OUT"
cd "$REPO"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"
timeout -k 10 400 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench rc=$?"; cut -c1-400 "$OUT/bench_default.json"
cd /tmp && export TMPDIR=/tmp
WEPP_SWEEP_UNFUSED=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/unfused" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --roofline-steps 0 > "$OUT/bench_unfused.json" 2> "$OUT/unfused.err"
echo "unfused rc=$?"
find "$OUT" -name "*.csv" -size +4M -delete
du -sh "$OUT"
