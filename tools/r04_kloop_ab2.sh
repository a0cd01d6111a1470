#!/bin/bash
# r04: K loop without epilogue (ablate 4) when the requesting waves (131072) or the store waves (262144) or both skip half of their
# MFMAs (timing only): whose chain is the pair's critical path?
set -e
mkdir -p gpurun_out
for rep in 1 2; do
  for ab in 4 131076 262148 393220; do
    echo "ablate=$ab" >> gpurun_out/r04_kloop2.log
    MOCR_GEMM_ABLATE=$ab timeout -k 10 200 python tools/gemm_bench.py enc 50432 t4096 >> gpurun_out/r04_kloop2.log 2>&1
  done
done
