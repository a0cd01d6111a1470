#!/bin/bash
set -o pipefail
O=gpurun_out/r3d
mkdir -p $O
for ab in 20 32 40; do
  MOCR_GEMM_ABLATE=$ab timeout -k 10 200 python tools/gemm_bench.py enc 50432 t4096 > $O/gemm_pers_ablate$ab.txt 2>&1; echo "ablate $ab rc=$?"; grep enc_ $O/gemm_pers_ablate$ab.txt
done
