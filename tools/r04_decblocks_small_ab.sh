#!/bin/bash
# r04: split-K block target (MOCR_DEC_BLOCKS, 150) for SMALL lone batches now that their GEMMs run on a four-slot ring
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0"
for b in 0 48 72 100 0 48 72 100; do
  MOCR_DEC_BLOCKS=$b MOCR_BENCH_ISOLATED=40,64,96,128,192,256 timeout -k 10 500 python bench.py --batch 256 --steps 4 --warmup 1 $X > gpurun_out/r04_decb_small_$b.$RANDOM.log 2>&1
done
