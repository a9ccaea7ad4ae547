#!/bin/bash
# gpurun -- 'bash tools/pmc_sq.sh <tag> [bench args]': SQ counters of one bench configuration (own pass, no trace domains)
set -u
TAG=${1:-sq}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --roofline-steps 0 $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d "$OUT/sq" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_sq.json" 2> "$OUT/sq.err"
echo "sq rc=$?"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD --output-format csv -d "$OUT/sq2" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_sq2.json" 2> "$OUT/sq2.err"
echo "sq2 rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for d in ("sq","sq2"):
    fs = glob.glob(sys.argv[1] + "/" + d + "/*/*_counter_collection.csv")
    if not fs: print("no csv for", d); continue
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(fs[0])):
        if "k_sweep" in row["Kernel_Name"]:
            agg[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    for k in sorted(agg): print("%-26s per-launch %.4g  (%d launches)" % (k, agg[k] / n[k], n[k]))
PY
find "$OUT" -name "*.csv" -size +2M -delete
