#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof_r03
mkdir -p $O
CMD="python bench.py --steps 10 --warmup 0 --lanes 1 --only-timed"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1; echo "trace rc=$?"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1; echo "write rc=$?"
python tools/blas_probe.py > $O/vendor_gemm.txt 2>&1
du -sh $O/*; tail -2 $O/trace.log
