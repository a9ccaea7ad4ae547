#!/bin/bash
# gpurun -- 'bash tools/r2_long.sh': parity tests, then the long-read bench line (configs[4] shape on one GPU)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_long
mkdir -p "$OUT"
cd "$REPO"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=4 > "$OUT/pytest.log" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -9 "$OUT/pytest.log"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --read-len 1200 --reads 200000 --steps 3 --warmup 1 --no-cpu-baseline --no-sensitivity --pcie-steps 0 --roofline-steps 0 > "$OUT/bench_long.json" 2> "$OUT/bench_long.err"
echo "bench rc=$?"; tail -2 "$OUT/bench_long.err"
python - <<'PY'
import json,os
b=json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r2_long/bench_long.json")))
print("long reads: value %.4g reads/s  ms/step %.1f  kernel_ms %.1f  setup %s" % (b["value"], b["ms_per_step"], b["roofline"]["kernel_ms_per_step"], b["config"]["setup_s"]))
print([ (s["nodes"], s["reads_routed"]) for s in b["streams"] if s["reads_routed"]])
PY
