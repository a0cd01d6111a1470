#!/bin/bash
# full GPU check of the working tree: every -m gpu test, then the default bench line
set -o pipefail
O=gpurun_out/r3h
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline --no-config4 --no-parity-leg > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3h/bench.json"))
e=d.get("encoder_only") or {}
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), "enc", round(e.get("kernel_ms",0),2), round(e.get("frac_of_mfma_peak",0),4), e.get("kernels"))
PY
