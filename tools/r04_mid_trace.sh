#!/bin/bash
# r04: kernel trace of ONE isolated batch of 1250 rows (a rank's share of the 8-GPU queue) and of 512 rows
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for b in 1250 512; do
  rm -rf gpurun_out/trace_b$b
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_b$b -- python tools/step_profile.py --batch $b --reps 2 > gpurun_out/r04_mid_b$b.log 2>&1
done
