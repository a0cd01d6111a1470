#!/bin/bash
set -o pipefail
O=gpurun_out/r3v
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "strip or folded_layernorm" > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
export MOCR_LIB=manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
for i in 1 2 3; do timeout -k 10 300 python tools/gemm_bench.py enc 50432 t409 2>&1 | grep -E "fc1" | grep -E "t4096|t4099"; done
