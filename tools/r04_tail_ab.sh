#!/bin/bash
# r04: the latent attention's last round.  2560 rows on 768 persistent blocks (three per CU) are 3.33 rounds: 256 blocks walk a fourth
# row while 512 idle.  A/B of the grid size (MOCR_LAT_BLOCKS, experiments build) and of row counts that are whole rounds.
set -e
export MOCR_LIB=$PWD/manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so
mkdir -p gpurun_out
for rep in 1 2; do
  for b in 0 640 512 704; do
    echo "blocks=$b" >> gpurun_out/r04_tail_lat.log
    MOCR_LAT_BLOCKS=$b N=2560 LS=100,197,300 timeout -k 10 120 python tools/latent_bench.py >> gpurun_out/r04_tail_lat.log 2>&1
  done
  echo "blocks=0 whole rounds" >> gpurun_out/r04_tail_lat.log
  N=2304,3072 LS=100,197,300 timeout -k 10 120 python tools/latent_bench.py >> gpurun_out/r04_tail_lat.log 2>&1
done
for rep in 1 2; do
  for b in 0 640; do
    MOCR_LAT_BLOCKS=$b timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --rows-per-rank-probe 0 > gpurun_out/r04_tail_bench_${b}_$rep.log 2>&1
  done
  # whole rounds per lane: 18 steps x 256 = 4608 = 2 x 2304; 24 steps = 6144 = 2 x 3072
  timeout -k 10 300 python bench.py --steps 18 --no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --rows-per-rank-probe 0 > gpurun_out/r04_tail_bench_s18_$rep.log 2>&1
  timeout -k 10 300 python bench.py --steps 24 --no-cpu-baseline --no-config4 --no-parity-leg --no-mixed --rows-per-rank-probe 0 > gpurun_out/r04_tail_bench_s24_$rep.log 2>&1
done
