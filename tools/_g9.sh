echo "== zeros t1024"; GEMM_BENCH_ZEROS=1 python tools/gemm_bench.py enc 50432 t1024 2>&1 | grep enc_
echo "== zeros t2048"; MOCR_GEMM_STAGGER=0 GEMM_BENCH_ZEROS=1 python tools/gemm_bench.py enc 50432 t2048 2>&1 | grep enc_
echo "== zeros t2048 no epilogue"; MOCR_GEMM_ABLATE=4 MOCR_GEMM_STAGGER=0 GEMM_BENCH_ZEROS=1 python tools/gemm_bench.py enc 50432 t2048 2>&1 | grep enc_
echo "== random t2048 (again)"; MOCR_GEMM_STAGGER=0 python tools/gemm_bench.py enc 50432 t2048 2>&1 | grep enc_
rocm-smi --showpower --showclocks 2>&1 | head -30
