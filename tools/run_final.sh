#!/bin/bash
set -o pipefail
O=gpurun_out/final_${ROUND:-r04}
mkdir -p $O
timeout -k 10 700 python bench.py --rows-per-rank-probe 1250 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 400 python bench.py --fp8-attention --no-cpu-baseline --no-config4 --no-parity-leg > $O/bench_fp8.json 2> $O/bench_fp8.err; echo "fp8 rc=$?"
timeout -k 10 400 python bench.py --queue 10000 --no-cpu-baseline --no-config4 --no-parity-leg --no-profile > $O/bench_queue10000.json 2> $O/bench_q.err; echo "queue rc=$?"
timeout -k 10 400 python bench.py --batch 64 --no-cpu-baseline --no-config4 --no-parity-leg > $O/bench_b64.json 2> $O/bench_b64.err; echo "b64 rc=$?"
python - <<'PY'
import json
for n in ("default","fp8","queue10000","b64"):
    d=json.load(open(f"gpurun_out/final_r04/bench_{n}.json"))
    e=d.get("encoder_only") or {}
    print(n, "value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), "enc", round(e.get("kernel_ms",0),2), round(e.get("frac_of_mfma_peak",0),4), "iso", d.get("isolated_step_ms"), "T32", d.get("regime_T32") and round(d["regime_T32"]["crops_per_s_this_rank"]))
PY
