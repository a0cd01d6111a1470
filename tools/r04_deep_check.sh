#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04_deep_tests.log 2>&1
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile"
for rep in 1 2; do
  MOCR_BENCH_ISOLATED=128,320,512,1024 timeout -k 10 300 python bench.py --rows-per-rank-probe 1250 $X > gpurun_out/r04_deep_final_$rep.log 2>&1
done
