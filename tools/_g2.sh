mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2b/gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2b/gputest.log
tail -15 gpurun_out/r2b/gputest.log
cp gpurun_out/parity_report.txt gpurun_out/r2b/parity_report.txt 2>/dev/null
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r2b/bench.json
