#!/bin/bash
# r04: four-slot ring also for 64 x 64-tile grids of up to 2 / 4 blocks per CU (two 64-KiB rings fit a CU)?
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0"
for m in 1 2 4 1 2 4; do
  MOCR_GEMM_DEEP_MULT64=$m MOCR_BENCH_ISOLATED=8,32,64,128,256,320,512,768 timeout -k 10 500 python bench.py --batch 768 --steps 3 --warmup 1 $X > gpurun_out/r04_deep64_$m.$RANDOM.log 2>&1
done
