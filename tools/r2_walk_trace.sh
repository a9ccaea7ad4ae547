#!/bin/bash
# gpurun -- 'bash tools/r2_walk_trace.sh': kernel trace of one probe leg with the walk on
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_walk_trace
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PROBE_LEGS=${PROBE_LEGS:-default}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/tools/walk_probe.py" > "$OUT/probe.log" 2> "$OUT/probe.err"
echo "rc=$?"; cut -c1-160 "$OUT/probe.log"
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("%-28s calls %4s avg %10.1f us  min %9.1f max %9.1f" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
find "$OUT" -name "*.csv" -size +2M -delete
