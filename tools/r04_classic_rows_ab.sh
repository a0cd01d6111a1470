#!/bin/bash
# r04: from how many rows does the latent attention (now three blocks per CU, 16-key tiles) beat the classic projected-K/V kernels
# for a batch submitted ALONE?  (r02 set the switch at 384 rows with the one-block-per-CU kernel.)  flags 64 = MOCR_FLAG_LATENT_ALWAYS.
set -e
mkdir -p gpurun_out
for f in 0 64 0 64; do
  MOCR_BENCH_ISOLATED=48,96,128,160,192,224,320,384,448 timeout -k 10 400 python bench.py --batch 512 --steps 4 --warmup 1 --no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --no-profile --rows-per-rank-probe 0 --engine-flags $f > gpurun_out/r04_classic_rows_$f.$RANDOM.log 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_classic_rows_*.log")):
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line)
            print(f.split("/")[-1], {int(k): round(v, 1) for k, v in sorted(d["isolated_step_ms"].items(), key=lambda kv: int(kv[0]))})
PY
