#!/bin/bash
# gpurun -- 'bash tools/r2_long_chunks.sh': long-read bench by the chunking of the window sweeps
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out/long_chunks
ARGS="--read-len 1200 --reads 200000 --steps 3 --warmup 1 --no-cpu-baseline --no-sensitivity --pcie-steps 0 --roofline-steps 0"
for cfg in "4096 1048576" "8192 1048576" "16384 1048576" "32768 1048576" "16384 262144" "65536 262144"; do
  set -- $cfg
  WEPP_TARGET_WAVES=$1 WEPP_CHUNK_BYTES=$2 timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('target_waves $1 chunk_bytes $2: %.4g reads/s %.1f ms' % (b['value'], b['ms_per_step']))"
done | tee gpurun_out/long_chunks/grid.txt
