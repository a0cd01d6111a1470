mkdir -p gpurun_out/r2t
rm -f gpurun_out/parity_report.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2t/gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2t/gputest.log; tail -6 gpurun_out/r2t/gputest.log
cp gpurun_out/parity_report.txt gpurun_out/r2t/parity_report.txt
python bench.py --queue 2000 --batch 256 --no-cpu-baseline --only-timed > gpurun_out/r2t/bench_queue.json 2> gpurun_out/r2t/bench_queue.err; echo "queue rc=$?"
MOCR_BENCH_FORCE_DIST=1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --queue 2000 --batch 256 --no-cpu-baseline --only-timed > gpurun_out/r2t/bench_queue_dist.json 2> gpurun_out/r2t/bench_queue_dist.err; echo "queue dist rc=$?"
MOCR_BENCH_FORCE_DIST=1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --only-timed > gpurun_out/r2t/bench_weak_dist.json 2> gpurun_out/r2t/bench_weak_dist.err; echo "weak dist rc=$?"
python - <<PY
import json
for n in ("bench_queue","bench_queue_dist","bench_weak_dist"):
    try:
        d=json.load(open(f"gpurun_out/r2t/{n}.json")); print(n, round(d["value"]), d["scaling"], d["steps"], round(d["ms_per_step"],2), d["config"]["rccl_world_size"])
    except Exception as e: print(n, "FAILED", e)
PY
