for st in 0 8 16 24 32 48 64; do
  echo "== stagger $st"
  MOCR_GEMM_STAGGER=$st python tools/gemm_bench.py enc 50432 t2048 2>&1 | grep enc_
done
echo "== default"
python tools/gemm_bench.py enc 50432 t2048 2>&1 | grep enc_
python tools/gemm_bench.py enc 403456 t2048 2>&1 | grep enc_
