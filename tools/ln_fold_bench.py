#!/usr/bin/env python3
"""GPU micro-benchmark of the LayerNorm-folding forms of the persistent encoder GEMM against the plain ones
(mocr_op_gemm / mocr_op_gemm_ln + HIP-event profile), batch-256 shapes.
    python tools/ln_fold_bench.py [M]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "manga-ocr_amd")]
_LAB = os.path.join(ROOT, "manga-ocr_amd", "manga_ocr", "_lib", "libmocr_hip_lab.so")
if "MOCR_LIB" not in os.environ and os.path.exists(_LAB):
    os.environ["MOCR_LIB"] = _LAB
import torch  # noqa: E402

from manga_ocr.engine import Engine  # noqa: E402
from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights  # noqa: E402


def timed(eng, fn, reps=5):
    for _ in range(2):
        fn()
    eng.profile_enable(True)
    eng.profile_reset()
    for _ in range(reps):
        fn()
    st = eng.profile_get()[0]
    eng.profile_enable(False)
    return st["total_ms"] / st["launches"] * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 50432
    eng = Engine(synthetic_weights(0), DEFAULT_SPEC, dtype="bf16", max_batch=8)
    Mp = (M + 255) // 256 * 256
    part = torch.zeros(Mp, 4, 2, device="cuda")
    for name, N, K, epi, tile in (("oproj", 768, 768, 3, 4096), ("fc2", 768, 3072, 3, 4096), ("qkv", 2304, 768, 1, 4096), ("fc1", 3072, 768, 2, 4096)):
        A = (torch.randn(Mp, K, device="cuda") * 0.5).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        csum = torch.randn(N, device="cuda")
        if epi == 3:
            out = torch.randn(Mp, N, device="cuda")
            xb = torch.zeros(Mp, N, device="cuda", dtype=torch.bfloat16)
            plain = timed(eng, lambda: eng.op_gemm(A, W, bias, out, out, M, N, K, epi, tile=tile, split_k=1))
            fold = timed(eng, lambda: eng.op_gemm_ln(A, W, bias, out, out, M, N, K, epi, tile, part, None, xb))
        else:
            out = torch.zeros(Mp, N, device="cuda", dtype=torch.bfloat16)
            part.normal_()
            part[:, :, 1] = part[:, :, 0].abs() * 1000 + 500
            plain = timed(eng, lambda: eng.op_gemm(A, W, bias, out, None, M, N, K, epi, tile=tile, split_k=1))
            fold = timed(eng, lambda: eng.op_gemm_ln(A, W, bias, out, None, M, N, K, epi, tile, part, csum, None))
        print(f"{name:6s} M{M} N{N} K{K}: plain {plain:7.1f} us   LayerNorm-folding form {fold:7.1f} us   ({fold - plain:+.1f})", flush=True)


if __name__ == "__main__":
    main()
