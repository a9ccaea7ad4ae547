#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
PROBE_LEGS="${LEGS:-default,k=4,k=8,p_n=0.02}" timeout -k 10 300 python tools/walk_probe.py 2>/dev/null | grep "walk=1" | cut -c1-100
