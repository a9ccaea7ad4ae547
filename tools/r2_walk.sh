#!/bin/bash
# gpurun -- 'bash tools/r2_walk.sh': parity tests with the walk on, then bench A/B (walk on / off)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2_walk
mkdir -p "$OUT"
cd "$REPO"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=5 > "$OUT/pytest.log" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -12 "$OUT/pytest.log"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python bench.py --no-cpu-baseline --pcie-steps 3 > "$OUT/bench_walk.json" 2> "$OUT/bench_walk.err"
echo "bench walk rc=$?"; tail -2 "$OUT/bench_walk.err"
timeout -k 10 500 python bench.py --no-walk --no-cpu-baseline --pcie-steps 3 > "$OUT/bench_sweep.json" 2> "$OUT/bench_sweep.err"
echo "bench sweep rc=$?"
python - <<'PY'
import json,os
for n in ("walk","sweep"):
    try:
        b=json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r2_walk/bench_%s.json"%n)))
    except Exception as e:
        print(n, "no json", e); continue
    print(n, "value %.4g ms/step %.3f kernel_ms %.3f pcie %.4g" % (b["value"], b["ms_per_step"], b["roofline"]["kernel_ms_per_step"], b["value_pcie_inclusive"] or 0))
    w=b.get("roofline_whole_tree")
    if w: print("   whole-tree %.4g reads/s identical %s" % (w["reads_per_s"], w["results_identical_to_timed_run"]))
    for s in b.get("sensitivity") or []:
        print("   %-20s %.4g reads/s  %.3f ms  mean_entries %.2f" % (s["reads"], s["reads_per_s"], s["ms_per_step"], s["mean_entries"]))
PY
