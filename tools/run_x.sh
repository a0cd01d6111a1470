#!/bin/bash
export MOCR_LIB=manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
for ab in 0 32768 65536 0 32768; do echo "ablate $ab"; MOCR_GEMM_ABLATE=$ab timeout -k 10 300 python tools/gemm_bench.py enc 50432 t4096 2>&1 | grep -E "t4096"; done
