mkdir -p gpurun_out/r2e
for a in 0 4 2 6 1 5 7; do
  echo "== ablate $a"
  MOCR_GEMM_ABLATE=$a python tools/gemm_bench.py enc 50432 t1024 2>&1 | grep -v Warning
done > gpurun_out/r2e/ablate.txt 2>&1
cat gpurun_out/r2e/ablate.txt
