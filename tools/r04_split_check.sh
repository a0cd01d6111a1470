#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_bf16_parity.py tests/test_gpu_compaction.py tests/test_gpu_product.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04_split_tests.log 2>&1
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile"
for rep in 1 2; do
  timeout -k 10 300 python bench.py --rows-per-rank-probe 1250 $X > gpurun_out/r04_split_probe_$rep.log 2>&1
done
timeout -k 10 300 python bench.py --queue 10000 $X --rows-per-rank-probe 0 > gpurun_out/r04_split_queue.log 2>&1
