echo "== outputs folded into 2048 rows"; MOCR_GEMM_ABLATE=32 MOCR_GEMM_STAGGER=0 python tools/gemm_bench.py enc 50432 t2048 2>&1 | grep enc_
echo "== outputs folded into 256 rows"; MOCR_GEMM_ABLATE=64 MOCR_GEMM_STAGGER=0 python tools/gemm_bench.py enc 50432 t2048 2>&1 | grep enc_
echo "== no epilogue"; MOCR_GEMM_ABLATE=4 MOCR_GEMM_STAGGER=0 python tools/gemm_bench.py enc 50432 t2048 2>&1 | grep enc_
