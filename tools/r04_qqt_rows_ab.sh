#!/bin/bash
# r04: from how many rows is the fused query kernel (dec_qqt_kernel, one launch) ahead of the two GEMM launches it replaces, now that
# batches of 257+ rows take the latent attention?  Experiments build; an isolated batch of each size.
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
for q in 1024 256 512 1024 256 512; do
  MOCR_DEC_QQT_ROWS=$q MOCR_BENCH_ISOLATED=320,384,512,640,768,896 timeout -k 10 400 python bench.py --batch 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0 > gpurun_out/r04_qqt_rows_$q.$RANDOM.log 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_qqt_rows_*.log")):
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line)
            print(f.split("/")[-1], {int(k): round(v, 1) for k, v in sorted(d["isolated_step_ms"].items(), key=lambda kv: int(kv[0]))})
PY
