#!/usr/bin/env python3
"""Measuring stick only (never linked by the product): the vendor GEMM (torch.matmul -> hipBLASLt) on the encoder's
four GEMM shapes, bf16 in / bf16 out, random operands, no epilogue.  python tools/blas_ref.py [M]"""
import sys

import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 50432
for name, N, K in (("qkv", 2304, 768), ("oproj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    A = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    for _ in range(3):
        C = A @ W.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        C = A @ W.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"vendor {name:6s} M={M} N={N} K={K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
