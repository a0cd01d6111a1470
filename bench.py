#!/usr/bin/env python3
"""Headline benchmark: manga crops/sec (224x224, max_len=300) on N MI355X.

A step = one pass of the hot path (encoder + greedy decode to max_len) over one batch of synthetic crops per GPU,
inputs already resident in HBM, outputs left in HBM, through the C ABI (mocr_recognize_device).  The default
workload is BASELINE configs[2]: batch = 256 crops per step (synthetic stand-ins for "real manga crops / full
manga-ocr-base weights", which are not obtainable offline: shapes, FLOPs and bytes are identical; with synthetic
weights EOS never fires, so every row decodes the full T = max_len - 1 = 299 steps - the worst case).

N > 1: one process per GPU (torch.distributed / RCCL), launched by torch.distributed.run.
  * weak scaling (default): every rank gets its own B crops per step, no data-path collective; the decoded ids are
    all-gathered ONCE per timed job (the only exchange step);
  * --queue Q (BASELINE configs[3], Q = 10000): strong scaling - ONE queue of Q crops is DEALT as the product's multi-GPU
    dispatcher deals it (manga_ocr.multi.deal_sizes: whole rounds of equal chunks of at most lanes x max_batch rows - 8 chunks
    of 1250 for 10,000 crops on 8 GPUs): the ranks pull chunk numbers from one shared counter (the reference's workers POP
    jobs, src/ui/main_window.py:4329-4335), decode them in batches of B, then ONE exchange of the [Q, max_len + 1] int32 rows
    (an all-reduce of a block every row of which exactly one rank wrote); value = Q / max-over-ranks wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--max-len L] [--queue Q] [--dtype bf16|fp32]

Prints ONE JSON line on rank 0.
"""
import argparse
import dataclasses
import json
import os
import subprocess
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
ROUND = "r04"

ENC_FLOPS_PER_CROP = 35_126_120_448


def dec_flops_per_crop(T):
    return 929_562_624 + 44_857_344 * T + 3_072 * T * (T + 1)


# which roofline bounds a launch: its algorithmic FLOP per byte against the ridge point of the part (2.5 PF / 8 TB/s = 312
# FLOP/B for bf16) - decided per launch from the numbers the engine reports, not from the kernel's name (at 2560 rows the
# decode-step FC1 sits at ~500 FLOP/B: above the ridge)
def bound_of(flops, nbytes, peak_tf):
    if flops <= 0 or nbytes <= 0:
        return "hbm"
    return "mfma" if flops / nbytes >= peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9) else "hbm"


def usable_cores():
    """Cores this process may actually run on: the affinity mask, cut by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        f = {k.strip(): v.strip() for k, v in (ln.split(":", 1) for ln in out.splitlines() if ":" in ln)}
        return f"{f.get('Model name', '?')}, {f.get('Socket(s)', '?')} socket(s) x {f.get('Core(s) per socket', '?')} cores, {f.get('CPU(s)', '?')} logical CPUs on the host"
    except (OSError, subprocess.SubprocessError):
        return "unknown CPU"


def cpu_baseline(args, weights):
    """The oracle (a port: plain torch fp32 restatement of the transformers path) on this host's cores, on a bounded
    sample of the same workload: B = 1 (the reference's calling pattern, src/ui/main_window.py:9801) and B = 8, at
    T = max_len - 1 and at T = 32.  `value` is the B = 8, T = max_len - 1 figure - the regime
    nearest to the GPU line's."""
    import torch
    from oracle.mocr_oracle import Oracle
    from manga_ocr.weights import DEFAULT_SPEC
    cores = usable_cores()
    torch.set_num_threads(cores)
    o = Oracle(weights, DEFAULT_SPEC)
    gray = np.random.RandomState(1234).randint(0, 256, size=(max(16, args.cpu_sample), 224, 224), dtype=np.uint8)
    o.recognize_ids(gray[:1], max_len=8)      # warm

    def rate(n, b, max_len):
        t0 = time.perf_counter()
        for i in range(0, n, b):
            o.recognize_ids(gray[i:i + b], max_len=max_len)
        return n / (time.perf_counter() - t0), time.perf_counter() - t0

    n8 = args.cpu_sample
    regimes = {}
    t_all = 0.0
    for name, n, b, ml in ((f"B=8,T={args.max_len - 1}", n8, 8, args.max_len), (f"B=1,T={args.max_len - 1}", max(2, n8 // 8), 1, args.max_len),
                           ("B=8,T=32", n8, 8, 33), ("B=1,T=32", max(4, n8 // 4), 1, 33)):
        r, dt = rate(n, b, ml)
        regimes[name] = {"crops_per_s": r, "crops": n, "seconds": dt}
        t_all += dt
    head = regimes[f"B=8,T={args.max_len - 1}"]
    return {"value": head["crops_per_s"], "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": f"{n8} synthetic 224x224 crops, batch 8, greedy decode to max_len={args.max_len} (T={args.max_len - 1}), torch fp32 eager, "
                      f"{cores} threads (every core this process may use) of: {cpu_model()}; all four regimes took {t_all:.1f} s",
            "regimes": regimes}


class _FakeEngine:
    """CPU stand-in for the HIP engine, ONLY for the gloo rehearsal of the N > 1 path (MOCR_BENCH_FAKE_ENGINE=1,
    tests/test_bench_gloo.py): the same submit / synchronize surface, ids[b] = (start, byte-sum of crop b mod 6000, eos).
    Never measured, never shipped: the product path fails loudly without the HIP library."""

    def __init__(self, max_len):
        self.max_len = max_len

    def recognize_device(self, d_gray, n, out_ids, out_len):
        import torch
        key = d_gray[:n].reshape(n, -1).to(torch.int64).sum(1) % 6000
        out_ids[:n].zero_()
        out_ids[:n, 0] = 2
        out_ids[:n, 1] = key.to(torch.int32)
        out_ids[:n, 2] = 3
        out_len[:n] = 3

    def synchronize(self):
        pass

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="crops per GPU per step (BASELINE configs[2]: 256; configs[1]: 64)")
    ap.add_argument("--max-len", type=int, default=300)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--lanes", type=int, default=2, help="internal batches the engine keeps in flight (streams + workspaces)")
    ap.add_argument("--max-batch", type=int, default=8192, help="rows of one internal engine batch: submitted steps are merged up to this")
    ap.add_argument("--queue", type=int, default=0, help="strong-scaling mode: ONE queue of this many crops dealt to the ranks in chunks (configs[3]: 10000)")
    ap.add_argument("--deal", default="equal", choices=("equal", "guided"),
                    help="--queue: the chunking policy of manga_ocr.multi.deal_sizes the ranks pull from (the product's default: equal)")
    ap.add_argument("--cpu-sample", type=int, default=48, help="crops per regime the CPU baseline (oracle) decodes: ~15-20 s of CPU work in all")
    ap.add_argument("--fp8-attention", action="store_true",
                    help="opt-in mode of BASELINE configs[4]: e4m3 key/value rows + fp8 MFMA in the decode attention (not the parity configuration)")
    ap.add_argument("--engine-flags", type=int, default=0, help="extra mocr_config.flags bits (A/B runs, e.g. 512 = MOCR_FLAG_NO_LN_FOLD)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config4", action="store_true", help="skip the variable-resolution + fp8-attention record")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-parity-leg", action="store_true", help="skip the fp32 parity-mode / bf16 id-match record (tests/golden crops)")
    ap.add_argument("--no-mixed", action="store_true", help="skip the mixed-lengths record (EOS-biased weights, row compaction on / off)")
    ap.add_argument("--mixed-steps", type=int, default=60, help="steps of B crops the mixed-lengths leg queues (a deeper queue lets a third lane start the next batch while two tail off)")
    ap.add_argument("--only-mixed", action="store_true", help="the mixed-lengths leg only (experiments: lanes / queue depth)")
    ap.add_argument("--rows-per-rank-probe", type=int, default=1250,
                    help="time ONE batch of this many rows (10000 / 8 = 1250: what a rank of the 8-GPU strong-scaling run decodes) and "
                         "report the strong-scaling bound it implies")
    ap.add_argument("--only-timed", action="store_true",
                    help="warm-up + timed region only (the rocprofv3 passes: every launch in the trace then belongs to the timed shape)")
    args = ap.parse_args()

    # ONE JSON line on stdout and nothing else: libraries write banners there (RCCL prints its version block on the
    # first communicator), so stdout is pointed at stderr until the line is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    from manga_ocr.engine import Engine
    from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 as: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    # MOCR_BENCH_FAKE_ENGINE=1: the rank / shard / gather / max-over-ranks logic of this file on CPU tensors over gloo, with a
    # stand-in engine - the world-2 rehearsal tests/test_bench_gloo.py runs where there is no GPU.  Timed region only.
    fake = bool(os.environ.get("MOCR_BENCH_FAKE_ENGINE"))
    dev = "cpu" if fake else "cuda"
    if fake:
        args.only_timed = True
        torch.cuda.synchronize = lambda *a, **k: None
    else:
        torch.cuda.set_device(local)
    dist = None
    # MOCR_BENCH_FORCE_DIST=1: take the collective path (process group, barrier, all-gather, max over ranks) even with
    # one rank - a rehearsal of the N > 1 code on a one-GPU box
    use_dist = world > 1 or bool(os.environ.get("MOCR_BENCH_FORCE_DIST"))
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if fake:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    spec = dataclasses.replace(DEFAULT_SPEC, max_len=args.max_len)
    weights = None if fake else synthetic_weights(0)
    args.max_batch = max(args.max_batch, args.batch)
    eng = _FakeEngine(args.max_len) if fake else Engine(weights, spec, dtype=args.dtype, device=local, max_batch=args.max_batch, lanes=args.lanes,
                                                         flags=(128 if args.fp8_attention else 0) | args.engine_flags)
    dtype_label = args.dtype + ("+fp8attn" if args.fp8_attention else "")
    B, L = args.batch, args.max_len
    strong = args.queue > 0
    chunks, store, deal_seq, dealt = [], None, [0], []
    if strong:
        # the queue is DEALT, as the product's multi-GPU dispatcher deals it (manga_ocr/multi.py: deal_sizes - whole rounds of
        # equal chunks of at most lanes x max_batch rows): the SPMD ranks pull chunk numbers from one shared counter (the
        # process group's store: an atomic add on rank 0's TCP store), decode what they pulled, and exchange once at the end
        from manga_ocr.multi import deal_sizes
        chunks = deal_sizes(args.queue, world, args.max_batch * args.lanes, policy=args.deal)
        steps_local = -(-max(b - a for a, b in chunks) // B)
        if use_dist:
            try:        # the process group's own store (rank 0's TCP store): `add` is an atomic fetch-and-add
                from torch.distributed import distributed_c10d
                store = distributed_c10d._get_default_store()
            except (ImportError, AttributeError):       # a torch without that accessor: a store of our own next to the group's port
                store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ.get("MASTER_PORT", "29500")) + 17, world, rank == 0)
    else:
        steps_local = 0
    # this rank's crops: global crop ids [rank*B, (rank+1)*B) of every step (weak); strong: every chunk replays the SAME B
    # crops from its first row on, whoever pulls it
    gray = np.random.RandomState(1234 + (0 if strong else rank)).randint(0, 256, size=(B, 224, 224), dtype=np.uint8)
    d_gray = torch.from_numpy(gray).to(dev)
    K = max(args.steps, args.warmup, steps_local, 1)
    d_ids = torch.zeros((K, B, L), dtype=torch.int32, device=dev)      # one output block per step
    d_len = torch.zeros((K, B), dtype=torch.int32, device=dev)
    if strong:
        d_all = torch.zeros((args.queue, L + 1), dtype=torch.int32, device=dev)      # every row of the queue: ids + length
    else:
        d_all = torch.zeros((world * K, B, L), dtype=torch.int32, device=dev) if use_dist else None
    torch.cuda.synchronize()

    def run(nsteps):
        # submit every step's batch (the engine merges them into fat internal batches and overlaps those on its
        # lanes), run them to completion, then the job's one exchange: all-gather of the decoded ids (RCCL over xGMI)
        if strong:
            key = f"mocr_bench_deal_{deal_seq[0]}"          # one counter per pass (every rank makes the same passes in the same order)
            deal_seq[0] += 1
            del dealt[:]
            d_all.zero_()
            nxt = 0
            while True:
                if store is not None:
                    idx = store.add(key, 1) - 1
                else:
                    idx, nxt = nxt, nxt + 1
                if idx >= len(chunks):
                    break
                lo, hi = chunks[idx]
                n, left = hi - lo, hi - lo
                ns = -(-n // B)
                for i in range(ns):
                    eng.recognize_device(d_gray, min(B, left), d_ids[i], d_len[i])
                    left -= B
                eng.synchronize()
                d_all[lo:hi, :L] = d_ids[:ns].reshape(-1, L)[:n]
                d_all[lo:hi, L] = d_len[:ns].reshape(-1)[:n]
                dealt.append(idx)
            if use_dist:
                dist.all_reduce(d_all)        # a row is written by exactly one rank (zeros elsewhere): the sum IS the gather
            return
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

    def timed(nsteps):
        fence()
        t0 = time.perf_counter()
        run(nsteps)
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    # ---- warm-up: W steps as asked, then ONE untimed pass of exactly the timed shape: the engine's decode graphs
    # are keyed by the row count of the merged batch, and that depends on how many steps are queued together - a
    # capture + instantiate must never sit inside the timed region
    if args.warmup > 0:
        run(args.warmup)
    if strong or args.warmup != args.steps:
        run(args.steps)
    dt = timed(args.steps)
    per_rank_chunks = [len(dealt)]
    if strong and use_dist:
        cnt = torch.tensor([len(dealt)], dtype=torch.int64, device=dev)
        allc = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allc, cnt)
        per_rank_chunks = [int(v) for v in allc.tolist()]
    crops_timed = args.queue if strong else world * B * args.steps
    value = crops_timed / dt
    steps_timed = max(steps_local, 1) if strong else args.steps

    extras = rank == 0 and not args.only_timed
    if args.only_mixed:
        args.no_profile = args.no_config4 = args.no_parity_leg = args.no_cpu_baseline = True
        args.rows_per_rank_probe = 0
    light = world > 1          # N > 1: the other ranks wait for rank 0 at the final barrier - keep its extra legs short
    # ---- T = 32 regime (max_len = 33: a typical speech bubble), same queue, same engine
    t32 = None
    if extras and not strong and not light and L > 33 and not args.only_mixed:
        eng.set_generate_max_length(33)
        run(args.steps)
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        d32 = time.perf_counter() - t0
        eng.set_generate_max_length(L)
        t32 = {"T": 32, "max_len": 33, "crops_per_s_this_rank": B * args.steps / d32, "ms_per_step": d32 / args.steps * 1e3}

    # ---- one isolated step (B crops submitted alone, nothing to merge with): the latency a single caller sees
    isolated = {}
    if extras and not light and not args.only_mixed:
        more = {int(v) for v in os.environ.get("MOCR_BENCH_ISOLATED", "").split(",") if v}      # extra sizes (experiments)
        for b in sorted({8, 64, 256, B} | more):
            if b > B:
                continue
            for rep in range(3):                      # the first call captures the decode graphs of this batch size
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                eng.recognize_device(d_gray, b, d_ids[0], d_len[0])
                eng.synchronize()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) * 1e3
            isolated[str(b)] = ms

    # ---- per-kernel durations, HIP events on the engine's stream, same workload (instrumented pass)
    roof, kernels, enc_only = None, [], None
    if extras and not args.no_profile:
        eng.profile_enable(True)
        eng.profile_reset()
        # ONE internal batch of the shape the timed region's lanes ran (the engine splits a queue over its idle lanes
        # while every part keeps >= 1280 rows: engine.hip pump_once SPLIT_MIN), alone on the GPU: durations free of overlap
        rows_q = min(args.max_batch * args.lanes, steps_timed * B)
        parts = max(1, min(args.lanes, rows_q // 1280))
        psteps = max(1, min(args.max_batch // B, -(-steps_timed // parts)))
        for i in range(psteps):
            eng.recognize_device(d_gray, B, d_ids[i % K], d_len[i % K])
        eng.synchronize()
        stats = eng.profile_get()
        tot = sum(s["total_ms"] for s in stats)
        peak_tf = MFMA_BF16_PEAK_TF if args.dtype == "bf16" else MFMA_F32_PEAK_TF
        for s in sorted(stats, key=lambda s: -s["total_ms"]):
            avg_ms = s["total_ms"] / s["launches"]
            mf = bound_of(s["flops"], s["bytes"], peak_tf) == "mfma"
            if mf:
                ach = s["flops"] / s["launches"] / (avg_ms * 1e-3) / 1e12
                peak, unit, alg = peak_tf, "TFLOP/s", s["flops"] / s["launches"]
            else:
                ach = s["bytes"] / s["launches"] / (avg_ms * 1e-3) / 1e9
                peak, unit, alg = HBM_PEAK_GBS, "GB/s", s["bytes"] / s["launches"]
            kernels.append({"kernel": s["name"], "launches": s["launches"], "avg_us": avg_ms * 1e3,
                            "share": s["total_ms"] / tot, "bound": "mfma" if mf else "hbm", "achieved": ach,
                            "peak": peak, "unit": unit, "frac": ach / peak, "algorithmic_per_launch": alg})
        k0 = kernels[0]
        rows_prof = psteps * B
        traffic = None
        try:   # HBM bytes per launch from the committed PMC passes (tools/summarize_profiles.py) of this row count
            pm = json.load(open(os.path.join(ROOT, "profiles", f"{ROUND}_pmc_traffic.json")))
            ent = pm.get("by_rows", {}).get(str(rows_prof), {}).get(k0["kernel"])
            if ent and args.dtype == "bf16" and L == 300:
                traffic = ent["traffic_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        roof = {"kernel": k0["kernel"], "bound": k0["bound"], "achieved": k0["achieved"], "peak": k0["peak"],
                "unit": k0["unit"], "frac": k0["frac"], "traffic": traffic,
                "traffic_source": (f"profiles/{ROUND}_pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                   "committed with the round - NOT measured inside this run") if traffic is not None else None,
                "avg_us": k0["avg_us"],
                "algorithmic_per_launch": k0["algorithmic_per_launch"], "rows": rows_prof, "share_of_step": k0["share"]}
        # ---- encoder only at this batch: the north star's ">= 50 % of the bf16 MFMA peak at batch 256" target
        eng.profile_reset()
        eng.encode(d_gray, B)
        est = eng.profile_get()
        eng.profile_enable(False)
        ems = sum(s["total_ms"] for s in est)
        enc_only = {"batch": B, "kernel_ms": ems, "crops_per_s": B / (ems * 1e-3),
                    "tflops": ENC_FLOPS_PER_CROP * B / (ems * 1e-3) / 1e12,
                    "frac_of_mfma_peak": ENC_FLOPS_PER_CROP * B / (ems * 1e-3) / 1e12 / peak_tf,
                    "note": "sum of the HIP-event durations of every encoder launch (patchify .. final LayerNorm), one pass",
                    "kernels": sorted(([s["name"], s["launches"], round(s["total_ms"] / s["launches"] * 1e3, 1),
                                        round(s["flops"] / max(s["total_ms"], 1e-9) / 1e9, 1)] for s in est), key=lambda r: -r[1] * r[2])}

    # ---- BASELINE configs[4] on this one GPU (its 8-GPU form is the driver's to launch): variable-resolution crops as
    # the application hands them over (host RGB arrays of any size) -> luminance + PIL-exact BILINEAR resize on the
    # device -> encoder -> greedy decode with the opt-in fp8 attention.  A second, smaller engine; host packing and the
    # H2D copy are inside the timed region (there is no device-resident form of a ragged crop list).
    cfg4 = None
    if extras and not strong and not light and not args.no_config4:
        rs = np.random.RandomState(4321)                       # h, w = round(exp(U(ln 32, ln 512)))
        n4, mb4 = 4096, 2048
        hw = np.rint(np.exp(rs.uniform(np.log(32), np.log(512), size=(n4, 2)))).astype(int)
        crops4 = [rs.randint(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in hw]
        eng4 = Engine(weights, spec, dtype=args.dtype, device=local, max_batch=mb4, lanes=2, flags=128)
        eng4.recognize_images(crops4)                          # warm: graphs, resample tables
        t0 = time.perf_counter()
        ids4, lens4 = eng4.recognize_images(crops4)
        d4 = time.perf_counter() - t0
        # the same engine, the same planes, device-resident (what the host path is held against): submitted like the
        # headline's steps, merged by the engine into the same 2 x 2048-row batches
        planes = np.concatenate([eng4.preprocess(crops4[i:i + 512]) for i in range(0, n4, 512)])
        dpl = torch.from_numpy(planes).cuda()
        o4 = torch.zeros((n4, L), dtype=torch.int32, device="cuda")
        l4 = torch.zeros((n4,), dtype=torch.int32, device="cuda")
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(0, n4, 256):
                eng4.recognize_device(dpl[i:i + 256], 256, o4[i:i + 256], l4[i:i + 256])
            eng4.synchronize()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        same = bool((o4.cpu().numpy() == ids4).all())
        eng4.close()
        del dpl, o4, l4
        cfg4 = {"workload": f"BASELINE configs[4], one GPU: {n4} variable-resolution RGB crops (32..512 px per side, RandomState(4321)) from host "
                            f"memory, luminance + PIL-exact resize on the device, {args.dtype} encoder, fp8-attention greedy decode to max_len={L}",
                "dtype": args.dtype + "+fp8attn", "crops_per_s": n4 / d4, "seconds": d4, "engine_max_batch": mb4, "lanes": 2,
                "includes": "host packing of the pixel rows, H2D copy, device preprocessing - pipelined: chunk k + 1 is prepared while chunk k decodes",
                "device_resident_crops_per_s_same_engine": n4 / best, "host_over_device_resident": best / d4,
                "ids_equal_device_resident": same, "mean_pixels_per_crop": float((hw[:, 0] * hw[:, 1]).mean())}

    # ---- what the north star's "bit-identical token ids" costs: the fp32 parity mode's rate on a 256-crop batch and its
    # id match against the reference ids (tests/golden/bf16_parity.npz: transformers' own greedy generate on these 256
    # crops), next to the benchmarked bf16 engine's rows-identical rate on the same crops
    parity = None
    gold_path = os.path.join(ROOT, "tests", "golden", "bf16_parity.npz")
    if extras and not strong and not light and not args.no_parity_leg and L == 300 and os.path.exists(gold_path):
        gold = np.load(gold_path)
        want, gaps = gold["ids_seed0"].astype(np.int32), gold["gaps_seed0"].astype(np.float32)
        gcrops = np.random.RandomState(777).randint(0, 256, size=(256, 224, 224), dtype=np.uint8)      # tests/gpu_util.crops(777, 256)
        dgc = torch.from_numpy(gcrops).cuda()
        o_ids = torch.zeros((256, L), dtype=torch.int32, device="cuda")
        o_len = torch.zeros((256,), dtype=torch.int32, device="cuda")

        def run_on(e):
            for _ in range(2):                       # the first pass captures the graphs
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                e.recognize_device(dgc, 256, o_ids, o_len)
                e.synchronize()
                torch.cuda.synchronize()
                dt_ = time.perf_counter() - t1
            got = o_ids.cpu().numpy()
            same_rows = (got == want).all(axis=1)
            worst = 0.0
            for b in np.nonzero(~same_rows)[0]:
                t = int(np.nonzero(got[b] != want[b])[0][0])
                worst = max(worst, float(gaps[b, t - 1]))
            return dt_, float(same_rows.mean()), float((got == want).mean()), worst

        dt_b, rows_b, toks_b, worst_b = run_on(eng)
        eng32 = Engine(weights, spec, dtype="fp32", device=local, max_batch=256, lanes=1)
        dt_f, rows_f, toks_f, worst_f = run_on(eng32)
        # where the parity mode's time goes: the encoder alone (HIP events over its launches), the rest is the 299 decode steps
        eng32.profile_enable(True)
        eng32.profile_reset()
        eng32.encode(dgc, 256)
        enc32_ms = sum(s_["total_ms"] for s_ in eng32.profile_get())
        eng32.profile_enable(False)
        eng32.close()
        parity = {"crops": "the 256 golden crops of tests/golden/bf16_parity.npz (reference ids: transformers greedy generate, fp32)",
                  "parity_mode_fp32": {"crops_per_s": 256 / dt_f, "ms_per_256_crop_batch": dt_f * 1e3, "ids_identical": toks_f,
                                       "rows_identical_frac": rows_f, "max_reference_margin_at_a_divergence": worst_f,
                                       "encoder_ms": enc32_ms, "decode_and_rest_ms": dt_f * 1e3 - enc32_ms,
                                       "encoder_tflops_f32_mfma": ENC_FLOPS_PER_CROP * 256 / (enc32_ms * 1e-3) / 1e12,
                                       "note": "f32-input MFMA (157 TF peak) for every GEMM, fp32 K/V caches (1.2 MB per crop and layer of "
                                               "cross keys/values per step): what bit-identical ids cost; DESIGN.md 2 has why split-bf16 "
                                               "GEMMs would not keep ids_identical at 1.0"},
                  "bf16_rows_identical_frac": rows_b, "bf16_tokens_identical_frac": toks_b,
                  "bf16_max_reference_margin_at_a_divergence": worst_b, "bf16_ms_per_256_crop_batch": dt_b * 1e3}

    # ---- rows of DIFFERENT lengths (r04).  In the reference every crop is its own generate() call and stops at its own EOS
    # (TF/generation/utils.py:2929-2937 via src/ui/main_window.py:9801); the headline's synthetic weights never emit EOS.  This leg
    # uses the EOS-biased synthetic weights of tests/golden/early_eos_seed1.npz (seed 1, eos_bias 1.1): rows end after ~17 .. 300
    # tokens.  Same queue shape as the headline (steps of B crops merged by the engine), with the engine's row compaction and -
    # second engine - without it (MOCR_FLAG_NO_COMPACTION); `mean_length_time_s` is the headline engine decoding the same
    # number of crops with EVERY row at the mixed queue's mean length (what an ideal scheduler's decode work amounts to).  The
    # queue is deep (--mixed-steps 60 = 15,360 crops): lanes take 2560 rows each as they fall idle.
    mixed = None
    if extras and not strong and not light and not args.no_mixed and args.dtype == "bf16" and L == 300:
        wm = synthetic_weights(1, eos_bias=1.1)
        mb = min(args.max_batch, 2560)
        K = max(K, 1)
        nq = args.mixed_steps * B
        gm = np.random.RandomState(4322).randint(0, 256, size=(nq, 224, 224), dtype=np.uint8)
        dgm = torch.from_numpy(gm).cuda()
        oi = torch.zeros((nq, L), dtype=torch.int32, device="cuda")
        ol = torch.zeros((nq,), dtype=torch.int32, device="cuda")

        def run_mixed(e):
            best, slots = 1e9, 0
            for _ in range(2):                       # the first pass captures the graphs of every compacted row count
                s0 = e.decode_slot_steps()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(0, nq, B):
                    e.recognize_device(dgm[i:i + B], min(B, nq - i), oi[i:i + B], ol[i:i + B])
                e.synchronize()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t1)
                slots = e.decode_slot_steps() - s0
            return best, slots, ol.cpu().numpy().copy(), oi.cpu().numpy().copy()

        res = {}
        # (a third lane starts the next batch of a deep queue while two tail off on a few hundred slots: the cheap form of
        # refilling finished slots from the queue)
        for name, fl, nl in (("compacted", 0, args.lanes), ("uncompacted", 2048, args.lanes), ("compacted_3_lanes", 0, max(3, args.lanes))):
            em = Engine(wm, spec, dtype=args.dtype, device=local, max_batch=mb, lanes=nl, flags=fl | args.engine_flags)
            dtm, slots, lens_m, ids_m = run_mixed(em)
            ncomp = em.compaction_count()
            em.close()
            res[name] = (dtm, slots, lens_m, ids_m, ncomp)
        lens_m = res["compacted"][2]
        tokens = int((lens_m - 1).sum())
        mean_len = float(lens_m.mean())
        # the headline engine, every row exactly mean_len tokens long (EOS never fires there)
        ml = int(round(mean_len))
        eng.set_generate_max_length(ml)
        tmean = 1e9
        for _ in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(0, nq, B):
                eng.recognize_device(d_gray, B, d_ids[(i // B) % K], d_len[(i // B) % K])
            eng.synchronize()
            torch.cuda.synchronize()
            tmean = min(tmean, time.perf_counter() - t1)
        eng.set_generate_max_length(L)
        mixed = {"workload": f"{nq} synthetic crops in steps of {B}, EOS-biased weights (seed 1, eos_bias 1.1), engine max_batch {mb}, {args.lanes} lanes, max_len {L}",
                 "lengths": {"min": int(lens_m.min()), "mean": mean_len, "max": int(lens_m.max()),
                             "deciles": [int(v) for v in np.percentile(lens_m, [10, 20, 30, 40, 50, 60, 70, 80, 90])]},
                 "crops_per_s": nq / res["compacted"][0], "seconds": res["compacted"][0], "compactions": res["compacted"][4],
                 "useful_token_fraction": tokens / max(res["compacted"][1], 1),
                 "uncompacted": {"crops_per_s": nq / res["uncompacted"][0], "seconds": res["uncompacted"][0],
                                 "useful_token_fraction": tokens / max(res["uncompacted"][1], 1)},
                 "three_lanes": {"crops_per_s": nq / res["compacted_3_lanes"][0], "seconds": res["compacted_3_lanes"][0],
                                 "time_over_mean_length_time": res["compacted_3_lanes"][0] / tmean,
                                 "ids_identical": bool((res["compacted_3_lanes"][3] == res["uncompacted"][3]).all())},
                 "ids_identical_to_uncompacted": bool((res["compacted"][3] == res["uncompacted"][3]).all()),
                 "mean_length_time_s": tmean, "mean_length_tokens": ml,
                 "time_over_mean_length_time": res["compacted"][0] / tmean}
        del dgm, oi, ol

    # ---- strong-scaling probe: ONE batch of the rows a rank of the 8-GPU queue run gets, alone on this GPU
    probe = None
    if extras and not light and args.rows_per_rank_probe > 0:
        rows = args.rows_per_rank_probe
        nst = -(-rows // B)
        for rep in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            left = rows
            for i in range(nst):
                eng.recognize_device(d_gray, min(B, left), d_ids[i % K], d_len[i % K])
                left -= B
            eng.synchronize()
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t1
        probe = {"rows": rows, "ms": dtp * 1e3, "crops_per_s_one_gpu": rows / dtp,
                 "extrapolated_8gpu_queue_crops_per_s_upper_bound": 8 * rows / dtp,
                 "extrapolated_speedup_over_this_run_upper_bound": (8 * rows / dtp) / value,
                 "note": "an EXTRAPOLATION, not a measurement: 8 x the rate of one rank's share (10000 / 8 rows) decoded alone on this one GPU - "
                         "an upper bound on what 8 ranks reach before start-up skew, the one all-gather and, in the product "
                         "(MangaOcr(devices=...)), the pull-based dealing of chunks; no 8-GPU run backs it until the driver's SCALE record does; "
                         "`this run` = the value of this JSON line"}

    cpu = None
    if extras and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, weights)

    if rank == 0:
        T = L - 1
        cfg_name = {256: "BASELINE configs[2]", 64: "BASELINE configs[1]"}.get(B, "custom batch")
        if strong:
            cfg_name = "BASELINE configs[3]" if args.queue == 10000 else "sharded queue"
            sizes = sorted({b - a for a, b in chunks}, reverse=True)
            workload = (f"{cfg_name}: ONE queue of {args.queue} synthetic 224x224 crops dealt to {world} GPU(s) in {len(chunks)} chunk(s) of "
                        f"{'/'.join(str(v) for v in sizes[:4])} rows (manga_ocr.multi.deal_sizes, policy {args.deal}: the product dispatcher's "
                        f"chunking; ranks pull chunk numbers from a shared counter), decoded in batches of {B}, ONE exchange of int32 [{args.queue}, {L + 1}] rows")
        else:
            workload = (f"{cfg_name}: batch={B} synthetic 224x224 crops per GPU per step (stand-ins for real manga crops: no dataset or "
                        f"checkpoint offline; same shapes, FLOPs and bytes)")
        out = {
            "metric": "manga crops/sec (224x224, max_len=300)", "value": value, "unit": "crops/s", "n_gpus": world,
            "steps": steps_timed, "warmup": args.warmup, "ms_per_step": dt / steps_timed * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": dtype_label, "data": "synthetic",
            "config": {"workload": workload + f", ViT-B/16 encoder + 2-layer BERT decoder, greedy decode max_len={L} (T={T} steps, EOS never fires "
                                   "with synthetic weights)",
                       "global_batch": world * B, "queue": args.queue or None, "engine_max_batch": args.max_batch, "lanes": args.lanes, "max_len": L,
                       "decode_steps": T, "parallelism": f"dp{world}", "rccl_world_size": world if use_dist else None,
                       "weights": "synthetic seed 0"},
            "algorithmic_gflop_per_crop": (ENC_FLOPS_PER_CROP + dec_flops_per_crop(T)) / 1e9,
            # `value` is the throughput of the whole queue of steps: the engine merges the submitted steps into internal
            # batches of up to engine_max_batch rows (split over its lanes).  One batch submitted ALONE takes (ms):
            "isolated_step_ms": isolated, "regime_T32": t32, "encoder_only": enc_only, "config4_variable_res_fp8": cfg4,
            "parity": parity, "mixed_lengths": mixed, "strong_scaling_probe": probe,
            "roofline": roof, "cpu_baseline": cpu, "kernels": kernels[:24],
        }
        if strong:
            out["dealing"] = {"policy": args.deal, "chunks": len(chunks), "chunk_rows": [b - a for a, b in chunks][:32],
                              "chunks_pulled_per_rank": per_rank_chunks, "counter": "c10d store add" if store is not None else "local (one rank)"}
        if fake:      # the rehearsal's evidence that every rank's rows arrived, in rank order
            out["data"] = "synthetic (FAKE ENGINE: a CPU rehearsal of the collective path, not a measurement)"
            out["gathered_checksum"] = int(d_all.to(torch.int64).sum().item()) if d_all is not None else None
            out["gathered_shape"] = list(d_all.shape) if d_all is not None else None
        line = json.dumps(out)
    else:
        line = None
    if use_dist:
        dist.barrier()           # rank 0's instrumented pass is over: every rank leaves together
        dist.destroy_process_group()
    eng.close()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
