mkdir -p gpurun_out/r2g
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "three_stage_ring and 2048" > gpurun_out/r2g/t.log 2>&1; tail -3 gpurun_out/r2g/t.log
echo "== new LDS epilogue"; MOCR_GEMM_STAGGER=0 python tools/gemm_bench.py enc 50432 t2048 2>&1 | grep enc_
echo "== old register epilogue"; MOCR_GEMM_STAGGER=0 MOCR_GEMM_ABLATE=16 python tools/gemm_bench.py enc 50432 t2048 2>&1 | grep enc_
echo "== new, fat"; MOCR_GEMM_STAGGER=0 python tools/gemm_bench.py enc 403456 t2048 2>&1 | grep enc_
echo "== encoder-only B=256 with 2048"; MOCR_GEMM_STAGGER=0 MOCR_ENC_TILE=2048 python tools/step_profile.py --batch 256 --encoder-only --reps 3
