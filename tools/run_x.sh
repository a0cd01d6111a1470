#!/bin/bash
set -o pipefail
O=gpurun_out/r3y
mkdir -p $O
export MOCR_LIB=manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4 --no-parity-leg > $O/$name.json 2> $O/$name.err; python - <<PY
import json
d=json.load(open("$O/$name.json"))
ks={k['kernel']:(round(k['avg_us'],1),k['launches']) for k in d['kernels']}
print("$name", round(d['value']), {k:ks[k] for k in ('gemm_dec_proj','dec_add_ln','gemm_dec_fc1','gemm_dec_fc2','gemm_dec_vocab','dec_qqt','gemm_dec_ctx','lat_attn_cross') if k in ks})
PY
}
run base MOCR_X=0
run tile64 MOCR_DEC_FAT_ROWS=100000
run blocks100 MOCR_DEC_BLOCKS=100
run blocks400 MOCR_DEC_BLOCKS=400
