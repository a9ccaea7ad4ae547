#!/bin/bash
# gpurun -- 'bash tools/r2_walk_occ2.sh': walk legs by stack rows (LDS per wave -> waves per SIMD) and job size
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out/walk_occ
for cfg in "16 32 32" "12 20 32" "12 20 128" "8 16 32" "10 18 64"; do
  set -- $cfg
  echo "== STACK8=$1 STACK16=$2 JOB_EVENTS=$3"
  WEPP_WALK_STACK8=$1 WEPP_WALK_STACK16=$2 WEPP_WALK_JOB_EVENTS=$3 PROBE_LEGS="default,k=4,k=8,p_n=0.02,p_n=0.05" timeout -k 10 400 python tools/walk_probe.py 2>/dev/null | grep "walk=1" | cut -c1-110
done | tee gpurun_out/walk_occ/grid2.txt
