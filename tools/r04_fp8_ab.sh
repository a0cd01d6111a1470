#!/bin/bash
# r04: the fp8 latent attention in the transposed-tile form (default) against r02's kernel (flag 1024 = MOCR_FLAG_LATENT_TILE32), one box
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fp8_attention.py -x -q -m gpu > gpurun_out/r04_fp8_tests.log 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_compaction.py -x -q -m gpu -k fp8 >> gpurun_out/r04_fp8_tests.log 2>&1
for f in 0 1024 0 1024; do
  timeout -k 10 300 python bench.py --fp8-attention --no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --rows-per-rank-probe 0 --engine-flags $f > gpurun_out/r04_fp8_bench_$f.$RANDOM.log 2>&1
done
