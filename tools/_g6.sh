mkdir -p gpurun_out/r2f
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "three_stage_ring and 2048" > gpurun_out/r2f/t.log 2>&1; tail -3 gpurun_out/r2f/t.log
for a in 0 4 6 5; do
  echo "== ablate $a"
  MOCR_GEMM_ABLATE=$a python tools/gemm_bench.py enc 50432 t2048 2>&1 | grep enc_
done > gpurun_out/r2f/ablate.txt 2>&1
cat gpurun_out/r2f/ablate.txt
python tools/gemm_bench.py enc 50432 t1024 2>&1 | grep enc_
python tools/gemm_bench.py enc 403456 t1024 2>&1 | grep enc_
python tools/gemm_bench.py enc 403456 t2048 2>&1 | grep enc_
