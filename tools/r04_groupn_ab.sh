#!/bin/bash
# r04: does the tile order of the persistent encoder GEMMs (column groups of N-tiles: the weight slices of a group stay in the
# XCD's L2 while the A row-panels stream through once per group) buy time or fabric traffic?  Experiments build, one box.
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
for rep in 1 2; do
  for g in -1 6 4 3; do
    MOCR_GEMM_GROUPN=$g timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --rows-per-rank-probe 0 > gpurun_out/r04_groupn_${g}_$rep.log 2>&1
  done
done
