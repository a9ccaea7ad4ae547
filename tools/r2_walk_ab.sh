#!/bin/bash
# gpurun -- 'bash tools/r2_walk_ab.sh': parity tests, then probe legs for a few walk thresholds
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_walk_ab
mkdir -p "$OUT"
cd "$REPO"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$OUT/pytest.log" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 "$OUT/pytest.log"
[ $rc -ne 0 ] && exit $rc
for thr in ${THRS:-8 16 48}; do
  echo "== WEPP_WALK_MAX_EVENTS=$thr"
  WEPP_WALK_MAX_EVENTS=$thr PROBE_LEGS="${LEGS:-default,k=2,k=4,k=8,p_n=0.02,p_n=0.05}" timeout -k 10 300 python tools/walk_probe.py 2> "$OUT/err_$thr.log" | cut -c1-130
done
