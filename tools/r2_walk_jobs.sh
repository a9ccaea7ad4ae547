#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
for je in 32 12 6; do
  echo "== WEPP_WALK_JOB_EVENTS=$je (max events 12)"
  WEPP_WALK_MAX_EVENTS=12 WEPP_WALK_JOB_EVENTS=$je PROBE_LEGS="default,k=4,k=8,p_n=0.02" timeout -k 10 300 python tools/walk_probe.py 2>/dev/null | grep "walk=1" | cut -c1-110
done
