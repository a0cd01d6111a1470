#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "latent" 2>&1 | tail -3
for i in 1 2; do
echo "new"; N=2560 LS=100,197,300 timeout -k 10 200 python tools/latent_bench.py 2>&1 | grep "n="
echo "old"; MOCR_LIB=manga-ocr_amd/manga_ocr/_lib/libmocr_hip_oldlat.so N=2560 LS=100,197,300 timeout -k 10 200 python tools/latent_bench.py 2>&1 | grep "n="
done
