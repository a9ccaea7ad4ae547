#!/bin/bash
# gpurun -- 'bash tools/r2_final_bench.sh': the three bench lines (default incl. PCIe leg, sensitivity and CPU baseline;
# 1.2 kb reads; whole-tree sweeps) with the roofline objects read from the committed counters of this build
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_final
mkdir -p "$OUT"; cd "$REPO"
timeout -k 10 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "default rc=$?"
timeout -k 10 400 python bench.py --read-len 1200 --reads 200000 --steps 5 --warmup 1 --no-cpu-baseline --no-sensitivity > "$OUT/bench_long_reads.json" 2> "$OUT/bench_long.err"; echo "long rc=$?"
timeout -k 10 400 python bench.py --no-crowns --no-walk --steps 3 --warmup 1 --no-cpu-baseline --no-sensitivity --pcie-steps 0 > "$OUT/bench_whole_tree.json" 2> "$OUT/bench_whole.err"; echo "whole rc=$?"
python3 - <<PY
import json
for n in ("default","long_reads","whole_tree"):
    b=json.load(open("$OUT/bench_%s.json"%n))
    print(n, "%.4g"%b["value"], b["unit"], "ms/step %.3f"%b["ms_per_step"], "pcie", b.get("value_pcie_inclusive"), "roofline", {k:b["roofline"].get(k) for k in ("bound","achieved","peak","frac","traffic")}, "cpu", (b.get("cpu_baseline") or {}).get("value"))
PY
