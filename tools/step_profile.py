#!/usr/bin/env python3
"""One isolated batch of B crops through the engine (nothing to merge with): wall time, and the per-kernel
breakdown from HIP events (an instrumented eager pass).  For the true kernel durations of the graph-replayed pass run it
under the profiler WITH THE INTERPRETER NAMED after `--` (the profiler's preloaded library initialises the GPU before the
program starts, so a `#!/usr/bin/env` hop - launching this file directly - would be an exec from a GPU-initialised process):

    python tools/step_profile.py --batch 256 [--max-len 300] [--flags N] [--reps 3] [--events]
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python tools/step_profile.py --batch 256
"""
import argparse
import dataclasses
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "manga-ocr_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--max-len", type=int, default=300)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--events", action="store_true", help="also print the HIP-event breakdown of an eager pass")
    ap.add_argument("--encoder-only", action="store_true")
    ap.add_argument("--dtype", default="bf16", help="bf16 or fp32 (the parity mode)")
    args = ap.parse_args()

    import torch
    from manga_ocr.engine import Engine
    from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights

    spec = dataclasses.replace(DEFAULT_SPEC, max_len=args.max_len)
    eng = Engine(synthetic_weights(0), spec, dtype=args.dtype, device=0, max_batch=args.batch, flags=args.flags, lanes=args.lanes)
    B, L = args.batch, args.max_len
    gray = np.random.RandomState(1234).randint(0, 256, size=(B, 224, 224), dtype=np.uint8)
    d_gray = torch.from_numpy(gray).cuda()
    d_ids = torch.zeros((B, L), dtype=torch.int32, device="cuda")
    d_len = torch.zeros((B,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    out = {"batch": B, "max_len": L, "flags": args.flags, "dtype": args.dtype}
    if args.encoder_only:
        eng.encode(d_gray, B)
        ts = []
        for _ in range(args.reps):
            eng.profile_enable(True); eng.profile_reset()
            eng.encode(d_gray, B)
            st = eng.profile_get(); eng.profile_enable(False)
            ts.append(sum(s["total_ms"] for s in st))
        out["encoder_kernel_ms"] = ts
        out["encoder_tflops"] = 35_126_120_448 * B / (min(ts) * 1e-3) / 1e12
        out["kernels"] = sorted(([s["name"], s["launches"], round(s["total_ms"] / s["launches"] * 1e3, 1),
                                  round(s["flops"] / max(s["total_ms"], 1e-9) / 1e9, 1)] for s in st), key=lambda r: -r[1] * r[2])
        print(json.dumps(out))
        eng.close()
        return
    ts = []
    for _ in range(args.reps + 1):
        t0 = time.perf_counter()
        eng.recognize_device(d_gray, B, d_ids, d_len)
        eng.synchronize()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    out["isolated_ms"] = [round(t, 2) for t in ts[1:]]
    out["first_call_ms"] = round(ts[0], 2)
    out["crops_per_s"] = B / (min(ts[1:]) * 1e-3)
    if args.events:
        eng.profile_enable(True); eng.profile_reset()
        eng.recognize_device(d_gray, B, d_ids, d_len)
        eng.synchronize()
        st = eng.profile_get(); eng.profile_enable(False)
        tot = sum(s["total_ms"] for s in st)
        out["event_total_ms"] = tot
        out["kernels"] = sorted(([s["name"], s["launches"], round(s["total_ms"] / s["launches"] * 1e3, 1), round(s["total_ms"], 2)]
                                 for s in st), key=lambda r: -r[3])
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
