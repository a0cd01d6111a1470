#!/bin/bash
export MOCR_LIB=manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "encoder_attention" 2>&1 | tail -3
for ab in 0 16 2560 5120 7680; do MOCR_ENC_ATTN_ABLATE=$ab python tools/enc_attn_bench.py 256 2>&1 | grep "impl 1" | sed "s/^/ablate $ab: /"; done
MOCR_ENC_ATTN_ABLATE=0 python tools/enc_attn_bench.py 256 2>&1 | grep "impl"
