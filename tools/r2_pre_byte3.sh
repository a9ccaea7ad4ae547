#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out/pre_byte
for rep in 1 2; do
for mn in 4000000000 8192 131072; do
  WEPP_IX_PRE_MIN_NODES=$mn timeout -k 10 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-sensitivity --roofline-steps 0 --pcie-steps 0 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('min_nodes $mn: %.4g reads/s  %.4f ms/step  kernel %.4f ms' % (b['value'], b['ms_per_step'], b['roofline']['kernel_ms_per_step']))"
done; done | tee gpurun_out/pre_byte/grid3.txt
