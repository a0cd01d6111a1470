mkdir -p gpurun_out/r2w
timeout -k 10 600 python -m pytest tests/test_gpu_fp8_attention.py -m gpu -x -q > gpurun_out/r2w/t.log 2>&1; tail -3 gpurun_out/r2w/t.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp8-attention > gpurun_out/r2w/bench_fp8.json 2> gpurun_out/r2w/bench_fp8.err; echo rc=$?
python - <<PY
import json
d=json.load(open("gpurun_out/r2w/bench_fp8.json"))
print(round(d["value"]), d["isolated_step_ms"])
for k in d["kernels"][:3]: print("  ", k["kernel"], k["launches"], round(k["avg_us"],1), round(k["share"],3), round(k["frac"],3))
PY
