#!/bin/bash
# gpurun -- 'bash tools/r2_pre_byte.sh': parity, then the walk legs with the per-entry pre-test byte off / on
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out/pre_byte
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pre_byte/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -2 gpurun_out/pre_byte/pytest.log
[ $rc -ne 0 ] && exit $rc
for mn in 4000000000 8192 0; do
  echo "== WEPP_IX_PRE_MIN_NODES=$mn"
  WEPP_IX_PRE_MIN_NODES=$mn PROBE_LEGS="default,k=4,k=8,p_n=0.02,p_n=0.05" timeout -k 10 400 python tools/walk_probe.py 2>/dev/null | grep "walk=1" | cut -c1-110
done | tee gpurun_out/pre_byte/grid.txt
