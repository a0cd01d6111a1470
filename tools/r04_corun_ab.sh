#!/bin/bash
# r04: can one lane's launch-bound GEMM chain run UNDER the other lane's HBM-bound attention launch?  The attention's three
# persistent blocks per CU take 156 KiB of LDS and 456 of 512 registers per SIMD lane: nothing of the other stream fits beside
# them.  Fewer attention blocks per CU (MOCR_LAT_BLOCKS) and decode GEMMs that fit the LDS left over (64 x 64 tiles: 32 KiB).
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0"
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py $X > gpurun_out/r04_corun_${name}.log 2>&1; }
for rep in 1 2; do
  run A_$rep MOCR_NOP=1
  run B_$rep MOCR_LAT_BLOCKS=512
  run C_$rep MOCR_LAT_BLOCKS=512 MOCR_DEC_TILE=64 MOCR_DEC_QQT_ROWS=0 MOCR_DEC_QTTILE=64
  run D_$rep MOCR_DEC_TILE=64 MOCR_DEC_QQT_ROWS=0 MOCR_DEC_QTTILE=64
  run E_$rep MOCR_LAT_BLOCKS=256
  run F_$rep MOCR_LAT_BLOCKS=256 MOCR_DEC_QQT_ROWS=0
  run G_$rep MOCR_LAT_BLOCKS=512 MOCR_DEC_TILE=64 MOCR_DEC_QQT_ROWS=0 MOCR_DEC_QTTILE=64 MOCR_DEC_BLOCKS=450
done
