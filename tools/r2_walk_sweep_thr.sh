#!/bin/bash
# gpurun -- 'bash tools/r2_walk_sweep_thr.sh': the walk's event threshold (reads with more events are swept)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_walk_thr
mkdir -p "$OUT"
cd "$REPO"
for thr in 16 48 128 512; do
  echo "== WEPP_WALK_MAX_EVENTS=$thr"
  WEPP_WALK_MAX_EVENTS=$thr PROBE_LEGS="default,k=2,k=4,p_n=0.02,p_n=0.05" timeout -k 10 300 python tools/walk_probe.py 2> "$OUT/err_$thr.log" | grep "walk=1" | cut -c1-150
done
