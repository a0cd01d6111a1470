#!/bin/bash
# r04: the latent attention launches are 142 us by the bench's HIP events (eager pass, an event pair around every launch) and 154 us
# in the rocprofv3 kernel trace of the graph-replayed run on the same box.  Trace of an EAGER run (MOCR_FLAG_NO_GRAPH = 2, no
# event markers) and of the graph run, back to back.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
for f in 0 2 0 2; do
  O=gpurun_out/evtrace_$f.$RANDOM
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python bench.py --steps 10 --warmup 0 --lanes 1 --only-timed --engine-flags $f > $O.log 2>&1; echo "flags $f rc=$?"
  grep -h "latent_attnT_kernel" $O/*/*kernel_stats.csv | cut -d, -f1-4,6,7
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --rows-per-rank-probe 0 > gpurun_out/evtrace_bench.log 2>&1
