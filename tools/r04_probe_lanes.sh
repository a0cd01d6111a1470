#!/bin/bash
# r04: one 1250-row batch (a rank's share of the 8-GPU queue) as one lane x 1250 rows or split over two / three idle lanes
set -e
mkdir -p gpurun_out
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --steps 4 --warmup 1"
for rep in 1 2; do
  for l in 1 2 3; do
    for r in 1250 2500 800; do
      timeout -k 10 300 python bench.py --lanes $l --rows-per-rank-probe $r $X > gpurun_out/r04_probe_l${l}_r${r}_$rep.log 2>&1
    done
  done
done
