#!/bin/bash
# gpurun -- 'bash tools/r2_trace.sh <tag> [bench args]': kernel trace + stats of a short bench run (no PMC passes)
set -u
TAG=${1:-t}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --roofline-steps 0 --no-sensitivity --pcie-steps 0 "$@" > "$OUT/bench.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT/kernel_stats.csv"; rm -rf "$OUT/trace"
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/kernel_stats.csv")))[:14]:
    print("%-72s calls=%-4s avg_us=%9.2f pct=%s" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
