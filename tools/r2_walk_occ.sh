#!/bin/bash
# gpurun -- 'bash tools/r2_walk_occ.sh': parity, then the walk legs at two job sizes
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out/walk_occ
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/walk_occ/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/walk_occ/pytest.log
[ $rc -ne 0 ] && exit $rc
for cfg in "16 32" "16 128" "16 256"; do
  set -- $cfg
  echo "== MAX_EVENTS=$1 JOB_EVENTS=$2"
  WEPP_WALK_MAX_EVENTS=$1 WEPP_WALK_JOB_EVENTS=$2 PROBE_LEGS="default,k=4,k=8,p_n=0.02,p_n=0.05" timeout -k 10 400 python tools/walk_probe.py 2>/dev/null | grep "walk=1" | cut -c1-110
done | tee gpurun_out/walk_occ/grid.txt
