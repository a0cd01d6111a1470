#!/bin/bash
# SQ counters of the persistent encoder GEMM (QKV shape, M = 50,432) with and without its DMA / epilogue (lab library).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/pmc_gemm
mkdir -p $O
export MOCR_LIB=manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
rocprofv3 -L > $O/counters.txt 2>&1
S1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU"
S2="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"
for ab in 0 4 6; do
  for s in 1 2; do
    if [ $s = 1 ]; then C="$S1"; else C="$S2"; fi
    MOCR_GEMM_ABLATE=$ab timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/ab${ab}_s$s -- python tools/gemm_bench.py enc 50432 "enc_qkv t4096" > $O/ab${ab}_s$s.log 2>&1; echo "ab $ab set $s rc=$?"
  done
done
python - <<'PY'
import csv, glob, collections
for ab in (0, 4, 6):
    agg = collections.defaultdict(list)
    for s in (1, 2):
        for f in glob.glob(f"gpurun_out/pmc_gemm/ab{ab}_s{s}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_pers_kernel" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("ablate", ab, {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
