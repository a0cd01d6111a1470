#!/usr/bin/env python3
"""Does one small batch finish sooner as several concurrent part-batches (one per lane)?  B rows submitted as P jobs of
B/P rows to an engine with P lanes (MOCR_SPLIT_MIN lowered so that every idle lane takes one job), against one job."""
import dataclasses, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "manga-ocr_amd")]
import numpy as np, torch
from manga_ocr.engine import Engine
from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights

L = 300
os.environ["MOCR_SPLIT_MIN"] = "4"      # read once by the library: every idle lane takes its share of whatever is queued
w = synthetic_weights(0)
for B in [int(v) for v in os.environ.get("BS", "8,16,64,128,256").split(",")]:
    gray = torch.from_numpy(np.random.RandomState(1).randint(0, 256, size=(B, 224, 224), dtype=np.uint8)).cuda()
    ids = torch.zeros((B, L), dtype=torch.int32, device="cuda")
    lens = torch.zeros((B,), dtype=torch.int32, device="cuda")
    ref = None
    for P in (1, 2, 4):
        if B % P or B // P < 4:
            continue
        eng = Engine(w, DEFAULT_SPEC, dtype="bf16", device=0, max_batch=B, lanes=P)
        n = B // P
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(P):
                eng.recognize_device(gray[k * n:], n, ids[k * n:], lens[k * n:])
            eng.synchronize()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
        got = ids.cpu().numpy().copy()
        if ref is None:
            ref = got
        print(f"B={B:4d} as {P} x {n:4d}: {ms:7.2f} ms   ids equal to the single batch: {bool((got == ref).all())}", flush=True)
        eng.close()
