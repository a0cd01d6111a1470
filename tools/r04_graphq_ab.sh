#!/bin/bash
# r04: row quantum of the decode graphs between 257 and 1024 rows (128): a 288-row batch decodes on 384 slots
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0"
for v in 128 64 128 64; do
  MOCR_GRAPH_Q_MID=$v MOCR_BENCH_ISOLATED=272,288,320,352,416,448,544,600,700 timeout -k 10 500 python bench.py --batch 768 --steps 4 --warmup 1 $X > gpurun_out/r04_graphq_$v.$RANDOM.log 2>&1
done
