#!/bin/bash
# r04: shorter decode chunks while rows are leaving?  MOCR_SMALL_CHUNK=2 / 4 with MOCR_SMALL_CHUNK_ROWS=100000 forces the chunk length of EVERY batch
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
for v in 0 2 0 2; do
  MOCR_SMALL_CHUNK=$v MOCR_SMALL_CHUNK_ROWS=100000 timeout -k 10 300 python bench.py --only-mixed > gpurun_out/r04_chunk_mixed_$v.$RANDOM.log 2>&1
done
