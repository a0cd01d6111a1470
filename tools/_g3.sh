mkdir -p gpurun_out/r2c
for g in -1 1 2 3 4 6; do
  MOCR_GEMM_GROUPN=$g python tools/step_profile.py --batch 256 --encoder-only --reps 3 > gpurun_out/r2c/enc256_g$g.json 2> gpurun_out/r2c/enc256_g$g.err
  echo "group $g"; python - <<PY
import json
d=json.load(open("gpurun_out/r2c/enc256_g$g.json"))
print(round(min(d["encoder_kernel_ms"]),2), round(d["encoder_tflops"]), [(k[0],k[2]) for k in d["kernels"][:6]])
PY
done
