#!/bin/bash
# gpurun -- 'bash tools/pmc_tlb.sh [bench args]': address-translation and L1 counters of the sweep (own pass, no trace domains)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_tlb
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1
grep -o "\b[A-Z0-9_]*UTCL[A-Z0-9_]*\b" "$OUT/avail.txt" | sort -u > "$OUT/utcl_names.txt"
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --roofline-steps 0 $*"
rocprofv3 --pmc TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_PENDING_STALL_CYCLES --output-format csv -d "$OUT/a" -- python3 "$REPO/bench.py" $ARGS > "$OUT/a.json" 2> "$OUT/a.err"
echo "a rc=$?"
rocprofv3 --pmc TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TA_TCP_STATE_READ TCP_TCC_READ_REQ_LATENCY --output-format csv -d "$OUT/b" -- python3 "$REPO/bench.py" $ARGS > "$OUT/b.json" 2> "$OUT/b.err"
echo "b rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for d in ("a","b"):
    fs = glob.glob(sys.argv[1] + "/" + d + "/*/*_counter_collection.csv")
    if not fs: print("no csv for", d); continue
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(fs[0])):
        if "k_sweep" in row["Kernel_Name"]:
            agg[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    for k in sorted(agg): print("%-30s per-launch %.4g  (%d launches)" % (k, agg[k] / n[k], n[k]))
PY
find "$OUT" -name "*.csv" -size +2M -delete
