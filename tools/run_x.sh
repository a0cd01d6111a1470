#!/bin/bash
set -o pipefail
export MOCR_LIB=manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
for i in 1 2; do
timeout -k 10 300 python tools/gemm_bench.py enc 50432 t4 2>&1 | grep -E "oproj|fc2" | grep -E "t4096|t4099|t4103"
done
