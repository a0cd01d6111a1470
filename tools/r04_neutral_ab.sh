#!/bin/bash
# r04: launch choices that do not touch the arithmetic made by the rows a compacted batch has LEFT instead of by its regime:
# MOCR_NEUTRAL_BY_ROWS=1 ring depth of the split-K GEMMs + rows per block of the fused query launch, =2 also the GEMM tile (the
# split over K stays the regime's).  ids must stay bit-identical (the compaction tests compare compacted with uncompacted engines).
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
MOCR_NEUTRAL_BY_ROWS=2 timeout -k 10 900 python -m pytest tests/test_gpu_compaction.py -x -q -m gpu > gpurun_out/r04_neutral_tests.log 2>&1
for v in 0 1 2 0 1 2; do
  MOCR_NEUTRAL_BY_ROWS=$v timeout -k 10 300 python bench.py --only-mixed > gpurun_out/r04_neutral_mixed_$v.$RANDOM.log 2>&1
done
