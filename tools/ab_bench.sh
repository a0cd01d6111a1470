#!/bin/bash
# A/B of experiment builds against the default library, alternating, REPS times (run on the GPU box from the repo root):
#     bash tools/ab_bench.sh "<bench.py flags>" REPS libA.so libB.so ...      (libraries under manga-ocr_amd/manga_ocr/_lib/)
# writes gpurun_out/ab_<name><rep>.log, one bench JSON line each ("def" = the default library).
FLAGS=$1; REPS=$2; shift 2
L=$PWD/manga-ocr_amd/manga_ocr/_lib
mkdir -p gpurun_out
for i in $(seq 1 "$REPS"); do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4 $FLAGS > gpurun_out/ab_def$i.log 2>&1 || exit 1
    for lib in "$@"; do
        n=$(basename "$lib" .so)
        MOCR_LIB=$L/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4 $FLAGS > gpurun_out/ab_$n$i.log 2>&1 || exit 1
    done
done
