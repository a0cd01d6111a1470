#!/bin/bash
# r04: compaction planned with the freshest unfinished-row count that has landed + half-length chunks while rows are leaving,
# against the library before the change (libmocr_hip_base.so): tests, then the mixed-lengths leg and the headline, alternating
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_compaction.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04_lag_tests.log 2>&1
L=$PWD/manga-ocr_amd/manga_ocr/_lib
for rep in 1 2; do
  for lib in libmocr_hip.so libmocr_hip_base.so; do
    MOCR_LIB=$L/$lib timeout -k 10 300 python bench.py --only-mixed > gpurun_out/r04_lag_mixed_${lib%.so}_$rep.log 2>&1
    MOCR_LIB=$L/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0 > gpurun_out/r04_lag_head_${lib%.so}_$rep.log 2>&1
  done
done
