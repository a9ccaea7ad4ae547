#!/bin/bash
# gpurun -- 'bash tools/profile_fitch.sh <tag> [bench_fitch args]': Fitch tests + kernel trace of tools/bench_fitch.py
set -u
TAG=${1:-fitch}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$REPO"
timeout -k 10 600 python -m pytest tests/test_fitch.py -m gpu -x -q > "$OUT/pytest.log" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 "$OUT/pytest.log"
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/tools/bench_fitch.py" "$@" > "$OUT/bench_fitch.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
cat "$OUT/bench_fitch.json"
head -8 "$OUT"/trace/*/*_kernel_stats.csv | cut -c1-200
find "$OUT" -name "*.csv" -size +2M -delete
