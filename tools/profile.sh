#!/bin/bash
# Profiles a program on the GPU box with rocprofv3; run as: gpurun -- 'bash tools/profile.sh <tag> [bench args]'
# Pass 1: kernel trace + stats.  Further passes: PMC counters, each group in its own run (no trace domains
# combined with --pmc; FETCH_SIZE and WRITE_SIZE do not fit one pass).  Summaries land in gpurun_out/prof_<tag>/;
# tools/summarize_profile.py condenses them into profiles/<name>/ and profiles/pmc_counters.json.
# PROFILE_PROG=tools/walk_probe.py (with PROBE_* in the environment) profiles a probe leg instead of bench.py;
# PROFILE_PASSES="fetch write sq ..." restricts the counter passes ("none": the kernel trace only);
# PROFILE_EXTRA="name:COUNTER COUNTER ...;name2:..." adds passes of other counters (e.g.
# "tlb:TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS;sqact:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS").
set -u
TAG=${1:-r3}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PROG=${PROFILE_PROG:-bench.py}
if [ "$PROG" = "bench.py" ]; then
  ARGS="--steps 2 --warmup 1 --no-cpu-baseline --roofline-steps 0 --no-sensitivity --pcie-steps 0 --no-legs $*"
else
  ARGS="$*"
fi
PASSES=${PROFILE_PASSES:-fetch write sq tcc ta tcp grbm}
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/$PROG" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rc=$?; echo "trace rc=$rc"; [ $rc -ne 0 ] && { tail -5 "$OUT/trace.err"; exit $rc; }
pass() {   # name counters...
  local name=$1; shift
  case " $PASSES " in *" $name "*) ;; *) return 0;; esac
  timeout -k 10 500 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 "$REPO/$PROG" $ARGS > "$OUT/bench_$name.json" 2> "$OUT/$name.err"
  local rc=$?; echo "$name rc=$rc"; [ $rc -ne 0 ] && { tail -5 "$OUT/$name.err"; exit $rc; }
  return 0
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD
pass tcc TCC_HIT_sum TCC_MISS_sum
pass ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass grbm GRBM_GUI_ACTIVE
if [ -n "${PROFILE_EXTRA:-}" ]; then
  IFS=';' read -ra EXTRA <<< "$PROFILE_EXTRA"
  for spec in "${EXTRA[@]}"; do
    name=${spec%%:*}; PASSES="$PASSES $name"
    pass "$name" ${spec#*:}
  done
fi
# keep only the small summaries (stats + per-kernel counter rows of our kernels)
find "$OUT" -name "*.csv" -size +12M -exec sh -c 'head -400 "$1" > "$1.head"; rm "$1"' _ {} \;
du -sh "$OUT"
