#!/bin/bash
# r04: A/B of the latent attention kernels on one box: flags 0 = latent_attnT_kernel (transposed score tile, two barriers per tile),
# 4096 = untransposed on 16-key tiles, 1024 = r03 shape (untransposed, 32-key tiles, one block per CU)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "latent_attention" > gpurun_out/r04_lat_tests.log 2>&1
for f in 0 4096 1024 0 4096 1024; do
  FLAGS=$f N=2560 LS=50,100,197,300 timeout -k 10 120 python tools/latent_bench.py >> gpurun_out/r04_lat_bench.log 2>&1
done
timeout -k 10 900 python -m pytest tests/test_gpu_compaction.py tests/test_gpu_bf16_parity.py -x -q -m gpu > gpurun_out/r04_compaction_tests.log 2>&1 || echo "compaction / bf16 parity tests FAILED" >> gpurun_out/r04_lat_bench.log
timeout -k 10 600 python -m pytest tests/test_gpu_hostile_stats.py -q -m gpu > gpurun_out/r04_hostile_tests.log 2>&1 || echo "hostile-statistics tests FAILED" >> gpurun_out/r04_lat_bench.log
for f in 0 4096 0 4096; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4 --no-parity-leg --engine-flags $f > gpurun_out/r04_bench_flags$f.$RANDOM.log 2>&1
done
