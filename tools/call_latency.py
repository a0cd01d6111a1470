#!/usr/bin/env python3
"""MangaOcr.__call__ as the application uses it (src/ui/main_window.py:9801): latency of one caller, and throughput of 15
worker threads calling one crop at a time (src/core/workers.py:209-247), for several batching windows.  Synthetic weights
never emit EOS, so generate(max_length) is set to 24 tokens to stand in for an ordinary speech-bubble text."""
import os, sys, time, statistics
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "manga-ocr_amd")]
import numpy as np
from PIL import Image
from manga_ocr import MangaOcr

rs = np.random.RandomState(5)
imgs = [Image.fromarray(rs.randint(0, 256, (int(rs.randint(60, 400)), int(rs.randint(60, 300)), 3), dtype=np.uint8), mode="RGB") for _ in range(64)]
for window in [float(v) for v in os.environ.get("WINDOWS", "2.0,0.3,0").split(",")]:
    m = MangaOcr(synthetic_seed=0, max_batch=64, lanes=1, batch_timeout_ms=window)
    m.engine.set_generate_max_length(24)
    for im in imgs[:8]:
        m(im)
    one = []
    for im in imgs[:48]:
        t0 = time.perf_counter(); m(im); one.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(15) as ex:
        list(ex.map(m, imgs * 8))
    dt = time.perf_counter() - t0
    print(f"batching window {window:4.1f} ms: one caller {statistics.median(one):6.2f} ms per call (min {min(one):.2f}); "
          f"15 threads {len(imgs) * 8 / dt:7.0f} crops/s", flush=True)
    m.close()
