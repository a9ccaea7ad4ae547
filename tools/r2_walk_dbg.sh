#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_walk_dbg; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for je in 32 8; do
  echo "== JOB_EVENTS=$je"
  WEPP_WALK_DEBUG=1 WEPP_WALK_MAX_EVENTS=16 WEPP_WALK_JOB_EVENTS=$je PROBE_LEGS=default timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t$je" -- python3 "$REPO/tools/walk_probe.py" 2>&1 | grep -E "walk=1|\[walk\]" | cut -c1-120
  f=$(find "$OUT/t$je" -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("   %-28s calls %4s avg %10.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
find "$OUT" -name "*.csv" -size +1M -delete
