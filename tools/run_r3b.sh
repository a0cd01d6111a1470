#!/bin/bash
set -o pipefail
O=gpurun_out/r3b
mkdir -p $O
timeout -k 10 120 tools/probe/bin/store_probe > $O/store_probe.txt 2>&1; echo "store_probe rc=$?"
for ab in 0 4 5 6; do
  MOCR_GEMM_ABLATE=$ab timeout -k 10 200 python tools/gemm_bench.py enc 50432 t2048 > $O/gemm_ablate$ab.txt 2>&1; echo "ablate $ab rc=$?"
done
cat $O/store_probe.txt; for ab in 0 4 5 6; do echo "ablate $ab"; grep enc_ $O/gemm_ablate$ab.txt; done
