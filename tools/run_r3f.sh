#!/bin/bash
set -o pipefail
O=gpurun_out/r3f
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --rows-per-rank-probe 1250 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3f/bench.json"))
e=d["encoder_only"]
print("value", round(d["value"]), "enc ms", round(e["kernel_ms"],2), "frac", round(e["frac_of_mfma_peak"],4), "T32", round(d["regime_T32"]["crops_per_s_this_rank"]))
print([k[:3] for k in e["kernels"][:7]])
print("cfg4", {k: (round(v,3) if isinstance(v,float) else v) for k,v in d["config4_variable_res_fp8"].items() if k not in ("workload","includes")})
print("parity", json.dumps(d["parity"])[:900])
print("probe", d["strong_scaling_probe"])
print("iso", d["isolated_step_ms"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
