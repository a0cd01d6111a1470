#!/bin/bash
# first GPU pass of round 3: store probe, vendor GEMM reference, current GEMM kernels, new parity tests
set -o pipefail
O=gpurun_out/r3a
mkdir -p $O
timeout -k 10 120 tools/probe/bin/store_probe > $O/store_probe.txt 2>&1; echo "store_probe rc=$?"
timeout -k 10 200 python tools/blas_ref.py 50432 > $O/blas_ref.txt 2>&1; echo "blas_ref rc=$?"
timeout -k 10 300 python tools/gemm_bench.py enc 50432 t2048 > $O/gemm_t2048.txt 2>&1; echo "gemm_bench rc=$?"
timeout -k 10 900 python -m pytest tests/test_gpu_bf16_parity.py tests/test_gpu_fp8_attention.py -x -q -m gpu -k "bench_shape or 2560_row or fp8_attention_engine or config4 or free_running" > $O/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.txt
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
cat $O/store_probe.txt $O/blas_ref.txt $O/gemm_t2048.txt
