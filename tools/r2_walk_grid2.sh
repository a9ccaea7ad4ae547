#!/bin/bash
# gpurun -- 'bash tools/r2_walk_grid2.sh': job-size grid of the chunked walks on the N-rich legs
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out/walk_grid2
for cfg in "16 32" "16 64" "16 128" "64 64" "16 16"; do
  set -- $cfg
  echo "== MAX_EVENTS=$1 JOB_EVENTS=$2"
  WEPP_WALK_MAX_EVENTS=$1 WEPP_WALK_JOB_EVENTS=$2 PROBE_LEGS="k=8,p_n=0.05" timeout -k 10 300 python tools/walk_probe.py 2>/dev/null | cut -c1-150
done | tee gpurun_out/walk_grid2/grid.txt
