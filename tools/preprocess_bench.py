#!/usr/bin/env python3
"""Variable-resolution crops (SURVEY.md §8(d) cfg 5: h, w = round(exp(U(ln 32, ln 512))), RandomState(4321)) through
the device preprocessing (mocr_preprocess: host packing + one H2D copy + L conversion + Pillow-exact resize), with
the per-kernel HIP-event times, and the oracle's numpy restatement / Pillow itself on the host beside it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "manga-ocr_amd")]
import numpy as np  # noqa: E402

from manga_ocr.engine import Engine  # noqa: E402
from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    rs = np.random.RandomState(4321)
    hw = np.rint(np.exp(rs.uniform(np.log(32), np.log(512), size=(n, 2)))).astype(int)
    imgs = [rs.randint(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in hw]
    src_mb = sum(im.nbytes for im in imgs) / 1e6
    eng = Engine(synthetic_weights(0), DEFAULT_SPEC, dtype="bf16", max_batch=8)
    eng.preprocess(imgs[:64])
    eng.profile_enable(True)
    eng.profile_reset()
    t0 = time.perf_counter()
    out = eng.preprocess(imgs)
    dt = time.perf_counter() - t0
    st = {s["name"]: s for s in eng.profile_get()}
    eng.profile_enable(False)
    kern_ms = sum(st[k]["total_ms"] for k in ("resize_h", "resize_v"))
    print(f"{n} crops, {src_mb:.0f} MB RGB: mocr_preprocess {dt * 1e3:.1f} ms wall = {n / dt:.0f} crops/s "
          f"(kernels {kern_ms:.2f} ms = {n / kern_ms * 1e3:.0f} crops/s, {src_mb / kern_ms:.1f} GB/s of source pixels)", flush=True)
    try:
        from PIL import Image
        t0 = time.perf_counter()
        ref = [np.asarray(Image.fromarray(im, mode="RGB").convert("L").resize((224, 224), Image.BILINEAR)) for im in imgs[:256]]
        dtp = time.perf_counter() - t0
        same = all((a == b).all() for a, b in zip(ref, out[:256]))
        print(f"Pillow on one host core: {256 / dtp:.0f} crops/s; first 256 planes identical to the device's: {same}")
    except ImportError:
        pass


if __name__ == "__main__":
    main()
