mkdir -p gpurun_out/r2d
run() { # name envs...
  name=$1; shift
  env "$@" python tools/step_profile.py --batch 256 --encoder-only --reps 3 > gpurun_out/r2d/$name.json 2> gpurun_out/r2d/$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2d/$name.json"))
print("$name", round(min(d["encoder_kernel_ms"]),2), round(d["encoder_tflops"]), [(k[0],k[2]) for k in d["kernels"][:6]])
PY
}
run base X=1
run all512 MOCR_ENC_TILE=512
run all256 MOCR_ENC_TILE=256
run all128 MOCR_ENC_TILE=128
run o512 MOCR_ENC_TILE_O=512 MOCR_ENC_TILE_FC2=512
run o256 MOCR_ENC_TILE_O=256 MOCR_ENC_TILE_FC2=256
run o128 MOCR_ENC_TILE_O=128 MOCR_ENC_TILE_FC2=128
