#!/bin/bash
# r04: with 64-row blocks, from how many rows does the fused query launch beat the two GEMM launches?  (MOCR_DEC_QQT_ROWS: 512)
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0"
for v in 512 257 384 512 257 384; do
  MOCR_DEC_QQT_ROWS=$v MOCR_BENCH_ISOLATED=288,320,384,448,512 timeout -k 10 500 python bench.py --batch 512 --steps 4 --warmup 1 $X > gpurun_out/r04_qqt_rows2_$v.$RANDOM.log 2>&1
done
