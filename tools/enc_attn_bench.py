#!/usr/bin/env python3
"""GPU micro-benchmark of the encoder attention kernel through the C ABI (mocr_op_enc_attention)."""
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    eng = Engine(synthetic_weights(0), DEFAULT_SPEC, dtype="bf16", max_batch=8)
    qkv = (torch.randn(n * 197 + 256, 2304, device="cuda") * 1.5).to(torch.bfloat16)
    ctx = torch.zeros(n * 197, 768, device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize()
    for impl in (1, 2):             # 1 = the product kernel, 2 = the r02 kernel (experiments build only)
        try:
            for _ in range(2):
                eng.op_enc_attention(qkv, ctx, n, impl)
        except Exception as exc:     # noqa: BLE001
            print(f"impl {impl}: {exc}")
            continue
        eng.profile_enable(True)
        eng.profile_reset()
        for _ in range(5):
            eng.op_enc_attention(qkv, ctx, n, impl)
        st = eng.profile_get()[0]
        eng.profile_enable(False)
        us = st["total_ms"] / st["launches"] * 1e3
        gb = n * 197 * (2304 + 768) * 2 / 1e9
        print(f"enc attention impl {impl} n={n}: {us:9.1f} us  {4.0 * 197 * 197 * 64 * 12 * n / us / 1e6:7.1f} TFLOP/s  {gb / us * 1e6:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
