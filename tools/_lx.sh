timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k latent 2>&1 | tail -2
N=4096 timeout -k 10 200 python tools/latent_bench.py 2>&1 | grep -v amdgpu.ids
