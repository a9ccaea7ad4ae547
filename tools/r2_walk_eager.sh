#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
for en in 0 3000 100000 1048576; do
  echo "== WEPP_WALK_EAGER_NODES=$en"
  WEPP_WALK_EAGER_NODES=$en WEPP_WALK_MAX_EVENTS=16 PROBE_LEGS="default,k=4,k=8" timeout -k 10 300 python tools/walk_probe.py 2>/dev/null | grep "walk=1" | cut -c1-110
done
