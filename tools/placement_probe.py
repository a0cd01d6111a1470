#!/usr/bin/env python3
"""Does the latent attention launch's speed depend on WHERE its key rows were allocated?  (r04: the cross launch reads 140 or 154 us
between processes / boxes with every other kernel equal.)  One process, the same launch on freshly allocated copies of the rows
(older copies kept alive, so every copy lands somewhere else), several alignments."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "manga-ocr_amd")]
import torch
from manga_ocr.engine import Engine
from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights
eng = Engine(synthetic_weights(0), DEFAULT_SPEC, dtype="bf16", max_batch=8)
n, L = 2560, 197
qt = (torch.randn(n, 16, 768, device="cuda") * 0.08).to(torch.bfloat16)
out = torch.zeros(n, 16, 768, device="cuda", dtype=torch.bfloat16)
keep = []
def run(x):
    for _ in range(2):
        eng.op_latent_attention(qt, x, out, n, L, L * 768)
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(20):
        eng.op_latent_attention(qt, x, out, n, L, L * 768)
    st = eng.profile_get()[0]; eng.profile_enable(False)
    return st["total_ms"] / st["launches"] * 1e3
for trial in range(10):
    pad = torch.empty((trial * 37 + 1) * 1024 * 1024 // 2, dtype=torch.bfloat16, device="cuda")      # shifts what follows
    x = torch.randn(n * L + 64, 768, device="cuda").to(torch.bfloat16)
    keep += [pad, x]
    print(f"trial {trial}: x at 0x{x.data_ptr():x} (mod 2 MiB = {x.data_ptr() % (2 << 20)}, mod 1 GiB = {(x.data_ptr() % (1 << 30)) >> 20} MiB): {run(x):7.1f} us   again {run(x):7.1f} us", flush=True)
# the same bytes at byte offsets inside one big block
big = torch.empty(n * L * 768 + 64 * 768 + (8 << 20), dtype=torch.bfloat16, device="cuda")
for off in (0, 64, 2048, 1 << 16, 1 << 19, 1 << 20):
    x = big[off:off + (n * L + 64) * 768].view(-1, 768)
    x.copy_(keep[1][: n * L + 64])
    print(f"offset {off * 2:8d} B: {run(x):7.1f} us", flush=True)
