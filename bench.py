#!/usr/bin/env python3
"""Headline benchmark: manga crops/sec (224x224, max_len=300) on N MI355X.

A step = one pass of the hot path (encoder + cross-K/V + greedy decode to max_len) over one batch
of synthetic crops per GPU, inputs already resident in HBM, outputs left in HBM, through the C ABI
(mocr_recognize_device).  N > 1: one process per GPU (torch.distributed / RCCL), the crop queue is
sharded across ranks with no data-path collective; the decoded token ids are all-gathered once per
step (the only exchange step).  Weak scaling: per-GPU batch is fixed.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--max-len L] [--dtype bf16|fp32]

Prints ONE JSON line on rank 0.  Synthetic weights (seed 0) and synthetic crops: no checkpoint or
dataset is reachable offline; FLOPs and bytes do not depend on the values, and with synthetic
weights EOS never fires, so every row decodes the full max_len-1 = 299 steps (the worst case).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "manga-ocr_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA
MFMA_F32_PEAK_TF = 157.3     # f32-input MFMA

ENC_FLOPS_PER_CROP = 35_126_120_448


def dec_flops_per_crop(T):
    return 929_562_624 + 44_857_344 * T + 3_072 * T * (T + 1)


# which roofline bounds each kernel class
MFMA_BOUND = ("gemm_enc_", "gemm_patch_embed", "gemm_cross_kv", "enc_attn_mfma")   # lat_attn_* / dec_* / layernorm: HBM


def cpu_baseline(args, weights):
    """The oracle (a port: plain torch fp32 restatement of the transformers path) on this host's
    cores, on a bounded sample of the same workload."""
    import torch
    from oracle.mocr_oracle import Oracle
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    from manga_ocr.weights import DEFAULT_SPEC
    o = Oracle(weights, DEFAULT_SPEC)
    n, b = args.cpu_sample, min(8, args.cpu_sample)
    gray = np.random.RandomState(1234).randint(0, 256, size=(n, 224, 224), dtype=np.uint8)
    o.recognize_ids(gray[:1], max_len=8)      # warm
    t0 = time.perf_counter()
    for i in range(0, n, b):
        o.recognize_ids(gray[i:i + b], max_len=args.max_len)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": f"{n} synthetic 224x224 crops, batch {b}, greedy decode to max_len={args.max_len} (T={args.max_len - 1}), "
                      f"torch fp32 eager, {cores} threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64, help="crops per GPU per step (BASELINE configs[1]: 64)")
    ap.add_argument("--max-len", type=int, default=300)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--lanes", type=int, default=2, help="internal batches the engine keeps in flight (streams + workspaces)")
    ap.add_argument("--max-batch", type=int, default=8192, help="rows of one internal engine batch: submitted steps are merged up to this")
    ap.add_argument("--cpu-sample", type=int, default=96, help="crops the CPU baseline (oracle) decodes: ~12 s on 16 host threads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    import torch
    from manga_ocr.engine import Engine
    from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights
    import dataclasses

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 as: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    torch.cuda.set_device(local)
    dist = None
    # MOCR_BENCH_FORCE_DIST=1: take the collective path (process group, barrier, all-gather, max over ranks) even with
    # one rank - a rehearsal of the N > 1 code on a one-GPU box
    use_dist = world > 1 or bool(os.environ.get("MOCR_BENCH_FORCE_DIST"))
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    spec = dataclasses.replace(DEFAULT_SPEC, max_len=args.max_len)
    weights = synthetic_weights(0)
    args.max_batch = max(args.max_batch, args.batch)
    eng = Engine(weights, spec, dtype=args.dtype, device=local, max_batch=args.max_batch, lanes=args.lanes)
    B, L = args.batch, args.max_len
    # this rank's shard of the crop queue: global crop ids [rank*B, (rank+1)*B) of every step
    gray = np.random.RandomState(1234 + rank).randint(0, 256, size=(B, 224, 224), dtype=np.uint8)
    d_gray = torch.from_numpy(gray).cuda()
    K = max(args.steps, args.warmup, 1)
    d_ids = torch.zeros((K, B, L), dtype=torch.int32, device="cuda")      # one output block per step
    d_len = torch.zeros((K, B), dtype=torch.int32, device="cuda")
    d_all = torch.zeros((world * K, B, L), dtype=torch.int32, device="cuda") if use_dist else None
    torch.cuda.synchronize()

    def run(nsteps):
        # submit every step's batch (the engine overlaps them on its lanes), run them to completion,
        # then the job's one exchange: all-gather of the decoded ids (RCCL over xGMI)
        for i in range(nsteps):
            eng.recognize_device(d_gray, B, d_ids[i], d_len[i])
        eng.synchronize()
        if use_dist:
            dist.all_gather_into_tensor(d_all, d_ids)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    value = world * B * args.steps / dt

    # ---- one isolated step (64 crops submitted alone, nothing to merge with): the latency a single caller sees
    isolated_ms = None
    if rank == 0:
        fence() if not use_dist else torch.cuda.synchronize()
        for rep in range(2):                      # the first call captures the decode graphs of this batch size
            t1 = time.perf_counter()
            eng.recognize_device(d_gray, B, d_ids[0], d_len[0])
            eng.synchronize()
            torch.cuda.synchronize()
            isolated_ms = (time.perf_counter() - t1) * 1e3

    # ---- per-kernel durations, HIP events on the engine's stream, same workload (instrumented pass)
    roof, kernels = None, []
    if rank == 0 and not args.no_profile:
        eng.profile_enable(True)
        eng.profile_reset()
        psteps = max(1, min(args.max_batch // B, args.steps))   # ONE merged internal batch: durations free of overlap
        for i in range(psteps):
            eng.recognize_device(d_gray, B, d_ids[i % K], d_len[i % K])
        eng.synchronize()
        stats = eng.profile_get()
        eng.profile_enable(False)
        tot = sum(s["total_ms"] for s in stats)
        peak_tf = MFMA_BF16_PEAK_TF if args.dtype == "bf16" else MFMA_F32_PEAK_TF
        for s in sorted(stats, key=lambda s: -s["total_ms"]):
            avg_ms = s["total_ms"] / s["launches"]
            mf = s["name"].startswith(MFMA_BOUND)
            if mf:
                ach = s["flops"] / s["launches"] / (avg_ms * 1e-3) / 1e12
                peak, unit = peak_tf, "TFLOP/s"
            else:
                ach = s["bytes"] / s["launches"] / (avg_ms * 1e-3) / 1e9
                peak, unit = HBM_PEAK_GBS, "GB/s"
            kernels.append({"kernel": s["name"], "launches_per_step": s["launches"] / psteps, "avg_us": avg_ms * 1e3,
                            "share": s["total_ms"] / tot, "bound": "mfma" if mf else "hbm", "achieved": ach,
                            "peak": peak, "unit": unit, "frac": ach / peak})
        k0 = kernels[0]
        traffic = None
        try:   # HBM bytes per launch from the committed PMC passes (tools/summarize_profiles.py), same config only
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            # the PMC passes were taken on one merged batch of pm["rows"] rows; both latent-attention launches
            # (self / cross) are the same kernel symbol, so the average is over both
            if k0["kernel"].startswith("lat_attn") and args.dtype == "bf16" and args.max_len == 300 and pm.get("rows") == psteps * B:
                hit = [v for k, v in pm["kernels"].items() if "latent_attn_kernel" in k]
                if hit:
                    traffic = sum(v["traffic_bytes"] * v["launches"] for v in hit) / sum(v["launches"] for v in hit)
        except (OSError, ValueError, KeyError):
            pass
        roof = {"kernel": k0["kernel"], "bound": k0["bound"], "achieved": k0["achieved"], "peak": k0["peak"],
                "unit": k0["unit"], "frac": k0["frac"], "traffic": traffic, "avg_us": k0["avg_us"], "share_of_step": k0["share"]}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, weights)

    if rank == 0:
        T = L - 1
        out = {
            "metric": "manga crops/sec (224x224, max_len=300)", "value": value, "unit": "crops/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: batch={B} synthetic 224x224 crops per GPU per step, ViT-B/16 encoder + "
                                   f"2-layer BERT decoder, greedy decode max_len={L} (T={T} steps, EOS never fires with synthetic weights)",
                       "global_batch": world * B, "engine_max_batch": args.max_batch, "lanes": args.lanes, "max_len": L, "decode_steps": T, "parallelism": f"dp{world}",
                       "weights": "synthetic seed 0"},
            "algorithmic_gflop_per_crop": (ENC_FLOPS_PER_CROP + dec_flops_per_crop(T)) / 1e9,
            # `value` is the throughput of the whole queue: the engine merges the submitted 64-crop steps into internal
            # batches of up to engine_max_batch rows.  One 64-crop step submitted alone takes:
            "isolated_step_ms": isolated_ms,
            "roofline": roof, "cpu_baseline": cpu, "kernels": kernels[:24],
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
