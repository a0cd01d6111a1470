#!/bin/bash
# r04: re-check: four-slot LDS ring (three K-tiles in flight) for gemm_kernel launches of at most one block per CU (MOCR_GEMM_DEEP=1)
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0"
for d in 0 1 0 1; do
  MOCR_GEMM_DEEP=$d timeout -k 10 300 python bench.py $X > gpurun_out/r04_deep_head_$d.$RANDOM.log 2>&1
  MOCR_GEMM_DEEP=$d MOCR_BENCH_ISOLATED=128,320,512,1024 timeout -k 10 500 python bench.py --batch 1024 --steps 3 --warmup 1 $X > gpurun_out/r04_deep_iso_$d.$RANDOM.log 2>&1
done
