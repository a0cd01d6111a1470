#!/bin/bash
# r04: re-check of r01 / r02 thresholds with this round's kernels: the block count a decode-step GEMM is split over K to reach
# (MOCR_DEC_BLOCKS: 150, 200 from 2048 rows), isolated batches and the headline
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0"
for b in 0 100 150 250 400 0; do
  MOCR_DEC_BLOCKS=$b MOCR_BENCH_ISOLATED=128,320,512,768,1024,1536 timeout -k 10 500 python bench.py --batch 2048 --steps 3 --warmup 1 $X > gpurun_out/r04_decblocks_iso_$b.$RANDOM.log 2>&1
  MOCR_DEC_BLOCKS=$b timeout -k 10 300 python bench.py $X > gpurun_out/r04_decblocks_head_$b.$RANDOM.log 2>&1
done
