#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out/pre_byte
for mn in 32768 131072 524288; do
  echo "== WEPP_IX_PRE_MIN_NODES=$mn"
  WEPP_IX_PRE_MIN_NODES=$mn PROBE_LEGS="default,k=8,p_n=0.02,p_n=0.05" timeout -k 10 400 python tools/walk_probe.py 2>/dev/null | grep "walk=1" | cut -c1-110
done | tee gpurun_out/pre_byte/grid2.txt
