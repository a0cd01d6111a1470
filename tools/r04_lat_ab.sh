#!/bin/bash
# r04: A/B of the latent attention's tile shape (16-key tiles on two blocks per CU against r03's 32-key tiles), one box;
# then the compaction tests
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "latent_attention" > gpurun_out/r04_lat_tests.log 2>&1
for f in 0 1024 0 1024; do
  FLAGS=$f N=2560 LS=50,100,197,300 timeout -k 10 120 python tools/latent_bench.py >> gpurun_out/r04_lat_bench.log 2>&1
done
timeout -k 10 900 python -m pytest tests/test_gpu_compaction.py -x -q -m gpu > gpurun_out/r04_compaction_tests.log 2>&1 || echo "compaction tests FAILED" >> gpurun_out/r04_lat_bench.log
for f in 0 1024 0 1024; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config4 --no-parity-leg --engine-flags $f > gpurun_out/r04_bench_flags$f.$RANDOM.log 2>&1
done
