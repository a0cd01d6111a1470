#!/bin/bash
# r04: with the fused query launch on 64-row blocks, from how many rows does the latent path beat the classic kernels for a batch alone?
# flags 64 = MOCR_FLAG_LATENT_ALWAYS; MOCR_DEC_QQT_ROWS=129: the fused launch from 129 rows
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0"
for f in 0 64 0 64; do
  MOCR_DEC_QQT_ROWS=129 MOCR_BENCH_ISOLATED=128,160,192,224,256 timeout -k 10 500 python bench.py --batch 256 --steps 4 --warmup 1 --engine-flags $f $X > gpurun_out/r04_classic2_$f.$RANDOM.log 2>&1
done
