#!/bin/bash
# final r03 numbers: the bench lines kept under profiles/ and the three rocprofv3 passes
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/run_final.sh
bash tools/profile_passes.sh gpurun_out/prof_r03 > gpurun_out/prof_r03.log 2>&1; echo "profile rc=$?"
tail -3 gpurun_out/prof_r03.log
