#!/bin/bash
export MOCR_LIB=manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
for ab in 0 4096 0 4096; do echo "ablate $ab"; MOCR_GEMM_ABLATE=$ab timeout -k 10 200 python tools/ln_fold_bench.py 2>&1 | grep -E "qkv|fc1"; done
