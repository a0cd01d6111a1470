#!/bin/bash
# final r03 numbers: the bench lines kept under profiles/ and the three rocprofv3 passes
set -o pipefail
rm -rf gpurun_out/prof_r03b
bash tools/run_final.sh
bash tools/profile_passes.sh gpurun_out/prof_r03b > gpurun_out/prof_r03b.log 2>&1; echo "profile rc=$?"
tail -3 gpurun_out/prof_r03b.log
