#!/bin/bash
# gpurun -- 'bash tools/r2_epp_jobs.sh': EPP placer at 16 M nodes by the number of jobs a call is cut into
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out/epp_jobs
for t in ${EPP_JOBS_GRID:-8192 32768 65536 131072 262144}; do
  WEPP_EPP_TARGET_JOBS=$t timeout -k 10 300 python tools/bench_epp.py --steps 2 --cpu-reads 0 --nodes 16000000 > gpurun_out/epp_jobs/b_$t.json 2>/dev/null
  python3 -c "
import json; b=json.load(open('gpurun_out/epp_jobs/b_$t.json')); p=b['phases']; print($t, 'jobs', p['jobs'], 'sweep1 %.1f sweep2 %.1f device %.1f ms' % (p['sweep1_ms'], p['sweep2_ms'], b['device_ms']))"
done | tee gpurun_out/epp_jobs/grid.txt
