#!/bin/bash
# Profiles tools/bench_epp.py with rocprofv3: gpurun -- 'bash tools/profile_epp.sh <tag> [bench_epp args]'
# Pass 1: kernel trace + stats.  Pass 2: SQ counters in their own run.
set -u
TAG=${1:-epp}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --cpu-reads 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/tools/bench_epp.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/tools/bench_epp.py" $ARGS > "$OUT/bench_sq.json" 2> "$OUT/sq.err"
echo "sq rc=$?"
cat "$OUT"/trace/*/*_kernel_stats.csv | cut -c1-200
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/pmc_sq/*/*_counter_collection.csv')
if f:
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        agg[r['Kernel_Name'][:40]][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items():
        if 'epp' in k: print(k, dict(v))
PY
find "$OUT" -name "*.csv" -size +2M -exec sh -c 'head -200 "$1" > "$1.head"; rm "$1"' _ {} \;
