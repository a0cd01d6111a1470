#!/bin/bash
set -o pipefail
O=gpurun_out/r3c
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "persistent or (three_stage and (4096 or 4097))" > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
for st in 0 8 16 32; do
MOCR_GEMM_STAGGER=$st timeout -k 10 300 python tools/gemm_bench.py enc 50432 t409 > $O/gemm_st$st.txt 2>&1; echo "stagger $st rc=$?"; grep enc_ $O/gemm_st$st.txt
done
MOCR_GEMM_ABLATE=4 timeout -k 10 300 python tools/gemm_bench.py enc 50432 t409 > $O/gemm_ab4.txt 2>&1; echo "ablate 4"; grep enc_ $O/gemm_ab4.txt
