#!/bin/bash
# r04: re-check: classic attention reads its K/V rows non-temporal from MOCR_ATTN_NT_ROWS rows (128)
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0"
for v in 128 32 100000 128 32 100000; do
  MOCR_ATTN_NT_ROWS=$v MOCR_BENCH_ISOLATED=40,64,96,128,160,192,224,256 timeout -k 10 500 python bench.py --batch 256 --steps 4 --warmup 1 $X > gpurun_out/r04_attn_nt_$v.$RANDOM.log 2>&1
done
