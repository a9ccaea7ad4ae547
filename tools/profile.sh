#!/bin/bash
# Profiles bench.py on the GPU box with rocprofv3; run as: gpurun -- 'bash tools/profile.sh <tag> [bench args]'
# Pass 1: kernel trace + stats.  Passes 2-4: PMC counters, each in its own run
# (no trace domains combined with --pmc).  Summaries land in gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --roofline-steps 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err"
echo "write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_sq.json" 2> "$OUT/sq.err"
echo "sq rc=$?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_tcc" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_tcc.json" 2> "$OUT/tcc.err"
echo "tcc rc=$?"
# keep only the small summaries (stats + per-kernel counter rows of our kernels)
find "$OUT" -name "*.csv" -size +2M -exec sh -c 'head -200 "$1" > "$1.head"; rm "$1"' _ {} \;
du -sh "$OUT"; find "$OUT" -type f | head -50
