#!/bin/bash
# r04: the fused query launch on 64-row blocks (16-KiB K-tiles, seven in flight, twice the blocks) up to MOCR_QQT_BM64_ROWS rows:
# never (0), up to 1280 rows (the default), always
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "fused or query or qqt or fat" > gpurun_out/r04_qqt_bm_tests.log 2>&1
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
X="--no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile"
for v in 0 1280 100000 0 1280 100000; do
  MOCR_QQT_BM64_ROWS=$v MOCR_BENCH_ISOLATED=512,768,1024,1280 timeout -k 10 500 python bench.py --batch 1280 --steps 4 --warmup 1 --rows-per-rank-probe 0 $X > gpurun_out/r04_qqt_bm_iso_$v.$RANDOM.log 2>&1
  MOCR_QQT_BM64_ROWS=$v timeout -k 10 300 python bench.py --rows-per-rank-probe 1250 $X > gpurun_out/r04_qqt_bm_head_$v.$RANDOM.log 2>&1
done
