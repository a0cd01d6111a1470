#!/usr/bin/env python3
"""Latency of one small batch whose rows END EARLY (EOS-biased synthetic weights): what a caller of the drop-in sees for
ordinary speech-bubble texts.  Prints the row lengths and the wall time of an isolated batch, for several batch sizes.
MOCR_SMALL_CHUNK / MOCR_SMALL_CHUNK_ROWS select the decode-chunk length of small batches (engine.hip, chunk_steps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "manga-ocr_amd")]
import numpy as np, torch
from manga_ocr.engine import Engine
from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights

bias = float(os.environ.get("EOS_BIAS", "1.1"))
w = synthetic_weights(1, eos_bias=bias)
for B in [int(v) for v in os.environ.get("BS", "1,8,16,64").split(",")]:
    eng = Engine(w, DEFAULT_SPEC, dtype="bf16", device=0, max_batch=B, lanes=1)
    gray = torch.from_numpy(np.random.RandomState(7).randint(0, 256, size=(B, 224, 224), dtype=np.uint8)).cuda()
    ids = torch.zeros((B, 300), dtype=torch.int32, device="cuda")
    lens = torch.zeros((B,), dtype=torch.int32, device="cuda")
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.recognize_device(gray, B, ids, lens)
        eng.synchronize()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ln = lens.cpu().numpy()
    print(f"B={B:3d}: lengths min {ln.min()} mean {ln.mean():.1f} max {ln.max()}   isolated batch {min(ts[1:]):7.2f} ms (median {np.median(ts[1:]):.2f})", flush=True)
    eng.close()
