#!/bin/bash
# Profiles bench.py on the GPU box with rocprofv3; run as: gpurun -- 'bash tools/profile.sh <tag> [bench args]'
# Pass 1: kernel trace + stats.  Further passes: PMC counters, each group in its own run (no trace domains
# combined with --pmc; FETCH_SIZE and WRITE_SIZE do not fit one pass).  Summaries land in gpurun_out/prof_<tag>/;
# tools/summarize_profile.py condenses them into profiles/<name>/ and profiles/pmc_counters.json.
set -u
TAG=${1:-r2}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --roofline-steps 0 --no-sensitivity --pcie-steps 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
pass() {   # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_$name.json" 2> "$OUT/$name.err"
  echo "$name rc=$?"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD
pass tcc TCC_HIT_sum TCC_MISS_sum
pass ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass grbm GRBM_GUI_ACTIVE
# keep only the small summaries (stats + per-kernel counter rows of our kernels)
find "$OUT" -name "*.csv" -size +2M -exec sh -c 'head -400 "$1" > "$1.head"; rm "$1"' _ {} \;
du -sh "$OUT"; find "$OUT" -type f | head -60
