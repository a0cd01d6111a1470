#!/usr/bin/env python3
"""GPU micro-benchmark of the GEMM kernel through the C ABI (mocr_op_gemm + HIP-event profile).
    python tools/gemm_bench.py            # a table of the recogniser's GEMM shapes
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "manga-ocr_amd")]
# the A/B tile codes (256 .. 2048, 4098) and the MOCR_* knobs live in the experiments build (build.py --experiments)
_LAB = os.path.join(ROOT, "manga-ocr_amd", "manga_ocr", "_lib", "libmocr_hip_lab.so")
if "MOCR_LIB" not in os.environ and os.path.exists(_LAB):
    os.environ["MOCR_LIB"] = _LAB
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

from manga_ocr.engine import Engine  # noqa: E402
from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights  # noqa: E402

EPI = {"slab": 0, "bias": 1, "gelu": 2, "resid": 3, "f32": 5}

SHAPES = [  # (name, M, N, K, epi, tile, split)
    ("dec_q t64", 2048, 768, 768, "bias", 64, 1), ("dec_q t128", 2048, 768, 768, "bias", 128, 1),
    ("dec_proj t64 s1", 2048, 768, 768, "slab", 64, 1), ("dec_proj t64 s2", 2048, 768, 768, "slab", 64, 2),
    ("dec_proj t128 s1", 2048, 768, 768, "slab", 128, 1), ("dec_proj t128 s2", 2048, 768, 768, "slab", 128, 2),
    ("dec_proj t128 s4", 2048, 768, 768, "slab", 128, 4),
    ("dec_fc1 t64 gelu", 2048, 3072, 768, "gelu", 64, 1), ("dec_fc1 t128 gelu", 2048, 3072, 768, "gelu", 128, 1),
    ("dec_fc2 t64 s1", 2048, 768, 3072, "slab", 64, 1), ("dec_fc2 t128 s1", 2048, 768, 3072, "slab", 128, 1),
    ("dec_fc2 t128 s2", 2048, 768, 3072, "slab", 128, 2), ("dec_fc2 t128 s4", 2048, 768, 3072, "slab", 128, 4),
    ("dec_vocab t64 s1", 2048, 6144, 768, "slab", 64, 1), ("dec_vocab t128 s1", 2048, 6144, 768, "slab", 128, 1),
    ("dec_vocab t256", 2048, 6144, 768, "f32", 256, 1),
]


def enc_shapes(M):
    out = []
    for name, N, K, epi in (("enc_qkv", 2304, 768, "bias"), ("enc_oproj", 768, 768, "resid"), ("enc_fc1", 3072, 768, "gelu"),
                            ("enc_fc2", 768, 3072, "resid")):
        for tile in (128, 256, 512, 1024, 2048, 4096, 4098, 4099, 4101, 4103, 4105):
            out.append((f"{name} t{tile}", M, N, K, epi, tile, 1))
    return out


def main():
    eng = Engine(synthetic_weights(0), DEFAULT_SPEC, dtype="bf16", max_batch=8)
    shapes = SHAPES
    if len(sys.argv) > 1 and sys.argv[1] == "enc":
        shapes = enc_shapes(int(sys.argv[2]) if len(sys.argv) > 2 else 2048 * 197)
        if len(sys.argv) > 3:
            shapes = [s for s in shapes if sys.argv[3] in s[0]]
    for name, M, N, K, epi, tile, split in shapes:
        Mp = (M + 255) // 256 * 256
        zero = bool(os.environ.get("GEMM_BENCH_ZEROS"))      # all-zero operands: the clock the chip holds is data-dependent
        A = (torch.randn(Mp, K, device="cuda") * (0.0 if zero else 0.5)).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * (0.0 if zero else 0.05)).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        out_f32 = epi in ("slab", "resid", "f32")
        out = torch.zeros((split if epi == "slab" else 1) * M, N, device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)
        resid = torch.randn(M, N, device="cuda") if epi == "resid" else None
        torch.cuda.synchronize()
        for _ in range(3):
            eng.op_gemm(A, W, bias, out, resid, M, N, K, EPI[epi], tile=tile, split_k=split)
        eng.profile_enable(True)
        eng.profile_reset()
        for _ in range(20 if M <= 4096 else 5):
            eng.op_gemm(A, W, bias, out, resid, M, N, K, EPI[epi], tile=tile, split_k=split)
        st = eng.profile_get()[0]
        eng.profile_enable(False)
        us = st["total_ms"] / st["launches"] * 1e3
        print(f"{name:22s} M{M:7d} N{N:5d} K{K:5d} {epi:5s} tile{tile:4d} split{split:3d}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
