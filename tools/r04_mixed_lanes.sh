#!/bin/bash
# r04: rows of different lengths with a DEEP queue - does a third / fourth lane, starting the next batch while two tail off, recover
# what the compacted tails leave idle?  (bench.py's mixed_lengths leg alone; 15,360 crops)
set -e
mkdir -p gpurun_out
for l in 2 3 4 2 3 4; do
  timeout -k 10 400 python bench.py --only-mixed --steps 2 --warmup 1 --mixed-steps 60 --lanes $l > gpurun_out/r04_mixed_lanes$l.$RANDOM.log 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_mixed_lanes*.log")):
    for line in open(f):
        if line.startswith("{"):
            m = json.loads(line)["mixed_lengths"]
            print(f.split("/")[-1], "compacted", round(m["crops_per_s"]), "uncompacted", round(m["uncompacted"]["crops_per_s"]),
                  "useful", round(m["useful_token_fraction"], 3), "x mean-length time", round(m["time_over_mean_length_time"], 3), "compactions", m["compactions"])
PY
