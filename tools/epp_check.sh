#!/bin/bash
# gpurun -- 'bash tools/epp_check.sh': EPP parity tests + the 16 M-node / 1 M-read measurement
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out/epp
timeout -k 10 600 python -m pytest tests/test_epp_gpu.py -m gpu -x -q > gpurun_out/epp/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/epp/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python tools/bench_epp.py --steps 2 --cpu-reads 0 "$@" > gpurun_out/epp/bench_epp.json 2> gpurun_out/epp/bench.err
echo "bench rc=$?"; python3 -c "
import json; b=json.load(open('gpurun_out/epp/bench_epp.json')); print({k:b[k] for k in ('reads_per_s_wall','device_ms','reads_per_s_device')}); print(b['phases'])"
