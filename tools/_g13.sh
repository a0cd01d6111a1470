mkdir -p gpurun_out/r2r
timeout -k 10 600 python -m pytest tests/test_gpu_fp8_attention.py -m gpu -x -q > gpurun_out/r2r/t.log 2>&1; tail -25 gpurun_out/r2r/t.log
grep -i "fp8" gpurun_out/parity_report.txt | tail -20
