#!/bin/bash
# r04: is the K loop of the persistent encoder GEMM bound by the requesting waves' serial chain (MFMAs + ~58 cycles per LDS-DMA
# request) while their SIMD partners wait at the barrier?  No-epilogue ablation (bit 4) of the pair loop with waves 0-3 requesting
# (tile code 4096) against every wave requesting its share (4105), and with a raised priority for the requesting waves (32768).
set -e
mkdir -p gpurun_out
for rep in 1 2; do
  for ab in 0 4 32772 65540; do
    echo "ablate=$ab" >> gpurun_out/r04_kloop.log
    MOCR_GEMM_ABLATE=$ab timeout -k 10 200 python tools/gemm_bench.py enc 50432 t4096 >> gpurun_out/r04_kloop.log 2>&1
    MOCR_GEMM_ABLATE=$ab timeout -k 10 200 python tools/gemm_bench.py enc 50432 t4105 >> gpurun_out/r04_kloop.log 2>&1
  done
done
