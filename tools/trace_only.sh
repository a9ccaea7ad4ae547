#!/bin/bash
# kernel-trace only: gpurun -- 'bash tools/trace_only.sh <tag> [bench args]'
set -u
TAG=${1:-t}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --roofline-steps 0 "$@" > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
cat "$OUT"/trace/*/*_kernel_stats.csv
python3 - "$OUT" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/trace/*/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_sweep' in r['Kernel_Name'] or 'k_route' in r['Kernel_Name'] or 'k_scatter' in r['Kernel_Name'] or 'k_finalize' in r['Kernel_Name']]
t0=None
for r in rows[-40:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if t0 is None: t0=s
    print(r['Kernel_Name'][:28].ljust(28), 'grid',r.get('Grid_Size_X','?').rjust(9),'lds',r['LDS_Block_Size'].rjust(6),'start %9.1f us dur %8.1f us'%((s-t0)/1e3,(e-s)/1e3))
PY
