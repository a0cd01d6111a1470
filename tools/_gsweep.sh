set -e
M=806912
echo "== base"; python tools/gemm_bench.py enc $M
for g in 4 6 9 12; do echo "== group_n $g"; MOCR_GEMM_GROUPN=$g python tools/gemm_bench.py enc $M; done
echo "== NT stores"; MOCR_GEMM_ABLATE=8 python tools/gemm_bench.py enc $M qkv
echo "== ablate 256 kernel qkv: 1 noMFMA"; MOCR_GEMM_ABLATE=1 python tools/gemm_bench.py enc $M "qkv t256"
echo "== 2 noDMA"; MOCR_GEMM_ABLATE=2 python tools/gemm_bench.py enc $M "qkv t256"
echo "== 4 noEpi"; MOCR_GEMM_ABLATE=4 python tools/gemm_bench.py enc $M "qkv t256"
echo "== 6 noDMA noEpi"; MOCR_GEMM_ABLATE=6 python tools/gemm_bench.py enc $M "qkv t256"
echo "== 5 noMFMA noEpi"; MOCR_GEMM_ABLATE=5 python tools/gemm_bench.py enc $M "qkv t256"
echo "== 3 noMFMA noDMA"; MOCR_GEMM_ABLATE=3 python tools/gemm_bench.py enc $M "qkv t256"
