#!/bin/bash
# gpurun -- 'bash tools/gpu_session.sh <step> [<step> ...]': the GPU-side steps of a working session, each with its own
# timeout, stopping at the first failure.  Steps:
#   parity      pytest tests/test_gpu_parity.py -m gpu            -> gpurun_out/session/pytest_parity.log
#   alltests    pytest tests -m gpu                               -> gpurun_out/session/pytest_gpu.log
#   bench       python bench.py (the driver's command)            -> gpurun_out/session/bench_default.json
#   probe       tools/walk_probe.py (PROBE_* from the environment) -> gpurun_out/session/probe.log
#   pcie        tools/pcie_rate.py: wepp_place_batch with host buffers, every pipeline depth   -> gpurun_out/session/pcie.log
#   probe@<v>   the same on variants/<v>/libwepp_place.so (tools/build_variant.sh)   -> gpurun_out/session/probe_<v>.log
#   probestats  the same on the -DWEPP_WALK_STATS build (variants/walkstats) with the walks' phase counters
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/session
mkdir -p "$OUT"
cd "$REPO"
for step in "$@"; do
  case $step in
    parity)   timeout -k 10 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s > "$OUT/pytest_parity.log" 2>&1; rc=$?; tail -4 "$OUT/pytest_parity.log";;
    alltests) timeout -k 10 1700 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; rc=$?; tail -4 "$OUT/pytest_gpu.log";;
    bench)    timeout -k 10 600 python bench.py ${BENCH_ARGS:-} > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; rc=$?; cut -c1-600 "$OUT/bench_default.json"; tail -3 "$OUT/bench_default.err";;
    probe)    timeout -k 10 900 python tools/walk_probe.py > "$OUT/probe${PROBE_TAG:-}.log" 2>&1; rc=$?; cut -c1-330 "$OUT/probe${PROBE_TAG:-}.log";;
    probestats) WEPP_PLACE_LIB=$REPO/variants/walkstats/libwepp_place.so WEPP_WALK_DEBUG=1 timeout -k 10 900 python tools/walk_probe.py > "$OUT/probestats.log" 2>&1; rc=$?; cut -c1-400 "$OUT/probestats.log";;
    pcie)     timeout -k 10 600 python tools/pcie_rate.py > "$OUT/pcie.log" 2>&1; rc=$?; grep -v amdgpu "$OUT/pcie.log" | cut -c1-200;;
    probe@*)  v=${step#probe@}; WEPP_PLACE_LIB=$REPO/variants/$v/libwepp_place.so timeout -k 10 900 python tools/walk_probe.py > "$OUT/probe_$v.log" 2>&1; rc=$?; cut -c1-200 "$OUT/probe_$v.log";;
    *) echo "unknown step $step"; rc=1;;
  esac
  echo "== $step rc=$rc"
  [ $rc -ne 0 ] && exit $rc
done
exit 0
