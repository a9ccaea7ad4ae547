#!/bin/bash
# gpurun -- 'bash tools/r2_walk_pmc.sh': SQ counters of the probe's kernels (walk on), one leg
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_walk_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PROBE_LEGS=${PROBE_LEGS:-default}
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc" -- python3 "$REPO/tools/walk_probe.py" > "$OUT/probe.log" 2> "$OUT/probe.err"
echo "rc=$?"
f=$(find "$OUT/pmc" -name "*counter_collection.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0][-30:]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="SQ_WAVES": n[k]+=1
for k in agg:
    a=agg[k]; c=max(n[k],1)
    if "walk" in k or "sweep" in k or "finalize" in k:
        print("%-32s launches %3d waves %9.0f VALU/launch %.3g SALU %.3g LDS %.3g  VALU/wave %.0f  wave_cycles(x4)/wave %.0f wait_any %.2f" % (k, c, a["SQ_WAVES"]/c, a["SQ_INSTS_VALU"]/c, a["SQ_INSTS_SALU"]/c, a["SQ_INSTS_LDS"]/c, a["SQ_INSTS_VALU"]/max(a["SQ_WAVES"],1), 4*a["SQ_WAVE_CYCLES"]/max(a["SQ_WAVES"],1), a["SQ_WAIT_ANY"]/max(a["SQ_WAVE_CYCLES"],1)))
PY
find "$OUT" -name "*.csv" -size +2M -delete
