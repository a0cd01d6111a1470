for a in 0 1 2 3 4; do echo "== ahead $a"; MOCR_LAT_AHEAD=$a N=4096 python tools/latent_bench.py 2>&1 | grep -v amdgpu.ids; done
MOCR_LAT_AHEAD=2 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k latent 2>&1 | tail -2
