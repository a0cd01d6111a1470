run() { echo "== $*"; env "$@" python bench.py --steps 64 --warmup 0 --lanes 1 --no-cpu-baseline > /tmp/b.json && python - <<PY
import json
d=json.load(open("/tmp/b.json"))
print("value", round(d["value"],1))
print(" ".join(f'{k["kernel"]}={k["avg_us"]:.1f}' for k in d["kernels"] if k["kernel"].startswith(("gemm_dec","dec_"))))
PY
}
run A=1
run MOCR_DEC_QTILE=128
run MOCR_DEC_QTTILE=128
run MOCR_DEC_BLOCKS=300
run MOCR_DEC_BLOCKS=600
run MOCR_DEC_TILE=64
