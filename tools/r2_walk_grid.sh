#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
for cfg in "16 32" "16 16" "16 8" "8 8" "8 4" "24 12"; do
  set -- $cfg
  echo "== MAX_EVENTS=$1 JOB_EVENTS=$2"
  WEPP_WALK_MAX_EVENTS=$1 WEPP_WALK_JOB_EVENTS=$2 PROBE_LEGS="default,k=4,p_n=0.02" timeout -k 10 300 python tools/walk_probe.py 2>/dev/null | grep "walk=1" | cut -c1-100
done
