#!/bin/bash
set -o pipefail
O=gpurun_out/r3u
mkdir -p $O
export MOCR_LIB=manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4 --no-parity-leg --steps 10 > $O/$name.json 2> $O/$name.err; python - <<PY
import json
d=json.load(open("$O/$name.json"))
e=d.get("encoder_only") or {}
print("$name", "value", round(d["value"]), "enc", round(e.get("kernel_ms",0),2), round(e.get("frac_of_mfma_peak",0),4), [(k[0][5:],k[2]) for k in e.get("kernels")][:6])
PY
}
run base MOCR_X=0
run ntx MOCR_GEMM_ABLATE=16384
run base2 MOCR_X=0
run ntx2 MOCR_GEMM_ABLATE=16384
