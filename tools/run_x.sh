#!/bin/bash
# End-of-round record on the GPU box: the bench lines kept under profiles/ (tools/run_final.sh) and the three rocprofv3
# passes (tools/profile_passes.sh).  Back in the build container:
#   python tools/summarize_profiles.py --round r03 --rows 2560 --stats gpurun_out/prof_r03b/trace \
#          --fetch gpurun_out/prof_r03b/fetch --write gpurun_out/prof_r03b/write --note "..."
#   cp gpurun_out/final_r03/bench_*.json profiles/   (as r03_bench_*.json)
set -o pipefail
rm -rf gpurun_out/prof_r03b
bash tools/run_final.sh
bash tools/profile_passes.sh gpurun_out/prof_r03b > gpurun_out/prof_r03b.log 2>&1; echo "profile rc=$?"
tail -3 gpurun_out/prof_r03b.log
