#!/bin/bash
# r04: re-check of an r02 threshold with this round's kernels: decode GEMMs on 128 x 128 tiles from MOCR_DEC_FAT_ROWS rows (512)
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
for f in 512 1024 1536 100000 512; do
  MOCR_DEC_FAT_ROWS=$f MOCR_BENCH_ISOLATED=320,512,640,768,1024,1280,1536,2048 timeout -k 10 500 python bench.py --batch 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0 > gpurun_out/r04_dectile_$f.$RANDOM.log 2>&1
done
