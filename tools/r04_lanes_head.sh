#!/bin/bash
# r04: the headline with two / three / four lanes (the idle-lane split gives each lane 2560 / 1707 / 1280 rows of the 5120-crop queue)
set -e
mkdir -p gpurun_out
for rep in 1 2; do
  for l in 2 3 4; do
    timeout -k 10 300 python bench.py --lanes $l --no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0 > gpurun_out/r04_lanes_head_${l}_$rep.log 2>&1
  done
done
