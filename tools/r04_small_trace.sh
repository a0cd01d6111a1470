#!/bin/bash
# r04: (1) the dealing paths on hardware: the two-children-on-one-card test and bench.py --queue through the collective path with a
# group of one; (2) kernel traces of ONE isolated batch of 64 / 256 rows (graph replay): durations and the gaps between launches
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 python -m pytest tests/test_gpu_product.py -x -q -m gpu -k "children or dispatcher" > gpurun_out/r04_deal_tests.log 2>&1
MOCR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --queue 10000 --no-cpu-baseline --no-profile > gpurun_out/r04_queue_dealt.log 2>&1
for b in 64 256; do
  rm -rf gpurun_out/trace_b$b
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_b$b -- python tools/step_profile.py --batch $b --reps 2 > gpurun_out/r04_small_b$b.log 2>&1
done
