#!/bin/bash
# The three rocprofv3 passes behind profiles/<round>_*.  Run on the GPU box from the repo root:
#     bash tools/profile_passes.sh gpurun_out/prof_r03
# then, back in the build container:
#     python tools/summarize_profiles.py --round r03 --rows 2560 --stats gpurun_out/prof_r03/trace \
#            --fetch gpurun_out/prof_r03/fetch --write gpurun_out/prof_r03/write --note "..."
# One lane, no warm-up shape, timed region only: every launch belongs to a 2560-row internal batch decoding alone.
# Counters are collected in passes of their own (FETCH_SIZE takes 3 of the 4 TCC slots), never together with a trace.
set -e
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python bench.py --steps 10 --warmup 0 --lanes 1 --only-timed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
du -sh "$OUT"/*
