#!/bin/bash
# r04: do two lanes still pay?  The same 5120-crop queue as ONE lane decoding 2560-row batches one after the other, as one
# 5120-row batch, and as the default two lanes x 2560 rows side by side.
set -e
mkdir -p gpurun_out
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0"
for rep in 1 2; do
  timeout -k 10 300 python bench.py $X > gpurun_out/r04_lane1_default_$rep.log 2>&1
  timeout -k 10 300 python bench.py --lanes 1 --max-batch 2560 $X > gpurun_out/r04_lane1_seq2560_$rep.log 2>&1
  timeout -k 10 300 python bench.py --lanes 1 $X > gpurun_out/r04_lane1_one5120_$rep.log 2>&1
  timeout -k 10 300 python bench.py --lanes 2 --max-batch 1280 $X > gpurun_out/r04_lane1_2x1280_$rep.log 2>&1
done
