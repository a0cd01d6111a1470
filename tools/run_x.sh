#!/bin/bash
set -o pipefail
O=gpurun_out/r3q
mkdir -p $O
MOCR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-parity-leg --no-profile > $O/dist_weak.json 2> $O/dist_weak.err; echo "weak rc=$?"
MOCR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout -k 10 300 python bench.py --gpus 1 --queue 2000 --no-cpu-baseline --no-config4 --no-parity-leg --no-profile > $O/dist_queue.json 2> $O/dist_queue.err; echo "queue rc=$?"
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-parity-leg --no-profile > $O/torchrun.json 2> $O/torchrun.err; echo "torchrun rc=$?"
python - <<'PY'
import json
for n in ("dist_weak","dist_queue","torchrun"):
    try:
        d=json.loads(open(f"gpurun_out/r3q/{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"]), d.get("scaling"), d.get("rccl_world_size"), d["n_gpus"])
    except Exception as ex: print(n, "ERR", ex)
PY
tail -3 $O/torchrun.err
