#!/bin/bash
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "encoder_attention" 2>&1 | tail -3
export MOCR_LIB=manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
for ab in 0 0 8 16; do MOCR_ENC_ATTN_ABLATE=$ab python tools/enc_attn_bench.py 256 2>&1 | grep "impl 1" | sed "s/^/ablate $ab: /"; done
