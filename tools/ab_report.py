#!/usr/bin/env python3
"""Print value / encoder_only / selected kernel averages of the gpurun_out/ab_*.log files written by tools/ab_bench.sh."""
import glob, json, os, sys
keys = sys.argv[1:] or ["gemm_enc_qkv", "gemm_enc_oproj", "gemm_enc_fc1", "gemm_enc_fc2", "enc_attn_mfma", "layernorm", "lat_attn_cross", "lat_attn_self"]
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "ab_*.log"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except (ValueError, IndexError):
        print(os.path.basename(f), "no JSON line"); continue
    k = {x["kernel"]: round(x["avg_us"], 1) for x in d.get("kernels") or []}
    enc = d.get("encoder_only") or {}
    print(f"{os.path.basename(f):40s} {d['value']:8.1f} crops/s  iso {({a: round(b, 1) for a, b in d['isolated_step_ms'].items()})}  enc {enc.get('kernel_ms') and round(enc['kernel_ms'], 2)} ms  "
          + " ".join(f"{n}={k[n]}" for n in keys if n in k))
