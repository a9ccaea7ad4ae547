#!/bin/bash
# gpurun -- 'bash tools/quick_gpu.sh': placement parity tests + a short bench line (no CPU baseline)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/quick
mkdir -p "$OUT"
cd "$REPO"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$OUT/pytest.log" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 "$OUT/pytest.log"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
python - <<'PY'
import json,os
b=json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/quick/bench.json")))
print("value %.4g reads/s  ms/step %.3f  sweep_ms %.3f  frac %.3f" % (b["value"], b["ms_per_step"], b["roofline"]["kernel_ms_per_step"], b["roofline"]["frac"]))
w=b.get("roofline_whole_tree")
if w: print("whole-tree: %.4g reads/s  kernel_ms %.1f  frac %.3f identical %s" % (w["reads_per_s"], w["kernel_ms_per_step"], w["frac"], w["results_identical_to_timed_run"]))
PY
