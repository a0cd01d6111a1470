#!/bin/bash
set -o pipefail
O=gpurun_out/r3e
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline --no-config4 > $O/bench_pers.json 2> $O/bench_pers.err; echo "bench rc=$?"
MOCR_ENC_BIG_TILE=2048 timeout -k 10 400 python bench.py --no-cpu-baseline --no-config4 > $O/bench_wide2.json 2> $O/bench_wide2.err; echo "bench rc=$?"
python - <<'PY'
import json
for n in ("pers","wide2"):
    d=json.load(open(f"gpurun_out/r3e/bench_{n}.json"))
    e=d["encoder_only"]
    print(n, "value", round(d["value"]), "enc ms", round(e["kernel_ms"],2), "frac", round(e["frac_of_mfma_peak"],4), "T32", round(d["regime_T32"]["crops_per_s_this_rank"]))
    print("   ", [k[:3] for k in e["kernels"][:7]])
PY
