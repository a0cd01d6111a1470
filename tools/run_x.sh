#!/bin/bash
set -o pipefail
O=gpurun_out/r3x2
mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4 --no-parity-leg --steps 10 "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
d=json.load(open("$O/$name.json"))
e=d.get("encoder_only") or {}
print("$name", "value", round(d["value"]), "enc", round(e.get("kernel_ms",0),2), round(e.get("frac_of_mfma_peak",0),4), [(k[0][5:],k[2]) for k in e.get("kernels")][:7])
PY
}
run fold
run nofold --engine-flags 512
run fold2
run nofold2 --engine-flags 512
