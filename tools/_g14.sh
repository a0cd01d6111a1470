mkdir -p gpurun_out/r2s
python bench.py --steps 20 --warmup 5 --fp8-attention --no-cpu-baseline > gpurun_out/r2s/bench_fp8.json 2> gpurun_out/r2s/bench_fp8.err; echo rc=$?
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2s/bench_bf16.json 2> gpurun_out/r2s/bench_bf16.err; echo rc=$?
python - <<PY
import json
for n in ("fp8","bf16"):
    d=json.load(open(f"gpurun_out/r2s/bench_{n}.json"))
    print(n, round(d["value"]), d["dtype"], d["isolated_step_ms"], d["regime_T32"]["crops_per_s_this_rank"])
    print("  roof", {k:(round(v,3) if isinstance(v,float) else v) for k,v in d["roofline"].items()})
    for k in d["kernels"][:8]: print("  ", k["kernel"], k["launches"], round(k["avg_us"],1), round(k["share"],3), round(k["frac"],3))
PY
