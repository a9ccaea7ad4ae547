#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_long_dbg
mkdir -p "$OUT"
cd "$REPO"
WEPP_DEBUG_PLANS=1 HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout -k 10 300 python bench.py --read-len 1200 --reads 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-sensitivity --pcie-steps 0 --roofline-steps 0 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "rc=$?"; grep -c "\[plan\]" "$OUT/bench.err"; grep "\[plan\]" "$OUT/bench.err" | head -40; grep -v "\[plan\]" "$OUT/bench.err" | tail -5
