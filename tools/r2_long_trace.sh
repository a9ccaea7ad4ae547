#!/bin/bash
# gpurun -- 'bash tools/r2_long_trace.sh': plan dump + kernel trace of the long-read bench
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_long_trace
mkdir -p "$OUT"
cd "$REPO"
ARGS="--read-len 1200 --reads 200000 --no-cpu-baseline --no-sensitivity --pcie-steps 0 --roofline-steps 0"
WEPP_DEBUG_PLANS=1 timeout -k 10 300 python bench.py $ARGS --steps 1 --warmup 0 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "rc=$?"; grep "\[plan\]" "$OUT/bench.err" | sort -u > "$OUT/plans.txt"; wc -l "$OUT/plans.txt"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python3 "$REPO/bench.py" $ARGS --steps 2 --warmup 1 > "$OUT/trace.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/trace/**/t_kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    k=(r["Kernel_Name"][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size",""), r.get("LDS_Block_Size",""))
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    agg[k][0]+=1; agg[k][1]+=d
out=sorted(agg.items(), key=lambda kv:-kv[1][1])
with open("$OUT/per_kernel.txt","w") as o:
    for k,(n,ms) in out[:60]:
        o.write("%-62s grid=%-10s lds=%-7s n=%-4d total_ms=%9.2f avg_ms=%8.3f\n"%(k[0],k[1],k[2],n,ms,ms/n))
print(open("$OUT/per_kernel.txt").read()[:6000])
PY
rm -rf "$OUT/trace"
