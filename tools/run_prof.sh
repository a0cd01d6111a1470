#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof_${ROUND:-r04}
mkdir -p $O
CMD="python bench.py --steps 10 --warmup 0 --lanes 1 --only-timed"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1; echo "trace rc=$?"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1; echo "write rc=$?"
# matrix-pipe utilisation: SQ counters in a pass of their own (never together with a trace)
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/sq -- $CMD > $O/sq.log 2>&1; echo "sq rc=$?"
python tools/blas_probe.py > $O/vendor_gemm.txt 2>&1
du -sh $O/*; tail -2 $O/trace.log
