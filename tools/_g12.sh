mkdir -p gpurun_out/r2p
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 10 --warmup 0 --lanes 1 --only-timed"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2p/trace -- $CMD > gpurun_out/r2p/trace.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2p/fetch -- $CMD > gpurun_out/r2p/fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2p/write -- $CMD > gpurun_out/r2p/write.log 2>&1
echo "write rc=$?"
du -sh gpurun_out/r2p/*
python bench.py --steps 20 --warmup 5 > gpurun_out/r2p/bench_default.json 2> gpurun_out/r2p/bench_default.err
echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r2p/bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','isolated_step_ms')}); print(d['roofline']); print({k:v for k,v in d['encoder_only'].items() if k!='kernels'})"
