#!/usr/bin/env python3
"""Micro-benchmark of the latent attention kernel (mocr_op_latent_attention), n rows x L keys."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "manga-ocr_amd")]
import torch
from manga_ocr.engine import Engine
from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights
FLAGS = int(os.environ.get("FLAGS", "0"))      # engine flags, e.g. 1024 = MOCR_FLAG_LATENT_TILE32
eng = Engine(synthetic_weights(0), DEFAULT_SPEC, dtype="bf16", max_batch=8, flags=FLAGS)
Ls = [int(v) for v in os.environ.get("LS", "32,96,197,300").split(",")]
for n in [int(v) for v in os.environ.get("N", "2048").split(",")]:
  for L in Ls:
      qt = (torch.randn(n, 16, 768, device="cuda") * 0.08).to(torch.bfloat16)
      x = torch.randn(n * L + 64, 768, device="cuda").to(torch.bfloat16)
      out = torch.zeros(n, 16, 768, device="cuda", dtype=torch.bfloat16)
      torch.cuda.synchronize()
      for _ in range(2):
          eng.op_latent_attention(qt, x, out, n, L, L * 768)
      eng.profile_enable(True); eng.profile_reset()
      for _ in range(10):
          eng.op_latent_attention(qt, x, out, n, L, L * 768)
      st = eng.profile_get()[0]; eng.profile_enable(False)
      us = st["total_ms"] / st["launches"] * 1e3
      tiles = (L + 31) // 32
      print(f"flags={FLAGS} n={n} L={L:4d}: {us:8.1f} us  {n * L * 1536 / us / 1e3:7.1f} GB/s   {us * 256 / n / tiles:6.2f} us per tile-block", flush=True)
