#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
for je in 32 8; do
  echo "== JOB_EVENTS=$je"
  WEPP_PLACE_LIB=$REPO/variants/walkstats/libwepp_place.so WEPP_WALK_DEBUG=1 WEPP_WALK_MAX_EVENTS=16 WEPP_WALK_JOB_EVENTS=$je PROBE_LEGS=${PROBE_LEGS:-default} timeout -k 10 300 python tools/walk_probe.py 2>&1 | grep -E "walk=1|walk stats" | cut -c1-200
done
