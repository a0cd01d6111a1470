#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/<dir>/*.csv) into the small files kept under profiles/.

    python tools/summarize_profiles.py --round r02 --rows 2560 --stats <dir> --fetch <dir> --write <dir> [--note ...]

All three passes run the SAME command (one lane, no warm-up shape, timed region only):
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python bench.py --steps 10 --warmup 0 --lanes 1 --only-timed
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python bench.py ...        (own pass: TCC has 4 slots, FETCH_SIZE takes 3)
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir> -- python bench.py ...
so every launch in them belongs to an internal batch of `--rows` rows decoding alone (kernel durations free of
overlap), the shape bench.py's instrumented pass measures with HIP events.

* <round>_kernel_stats.csv   : the rocprofv3 --stats summary, verbatim
* <round>_kernel_classes.json: per kernel CLASS as bench.py names it (the latent attention is two instantiations,
                               latent_attn_kernel<true> = self, <false> = cross): calls, avg / min / max ns
* <round>_pmc_traffic.json   : by_rows[rows][class] = FETCH_SIZE / WRITE_SIZE per launch and the corrected HBM bytes
      traffic_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950: FETCH_SIZE counts half of a wide
      16 B/lane read stream; WRITE_SIZE is exact - MI355X_MICROARCH.md, HBM section)
  next to the algorithmic bytes per launch, so that roofline.frac = algorithmic / avg_ns / 8 TB/s and
  traffic / algorithmic can be recomputed from these files alone.
"""
import argparse
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLASSES = [  # (substring of the kernel symbol, bench.py's class name)
    ("latent_attnT_kernel<true", "lat_attn_self"), ("latent_attnT_kernel<false", "lat_attn_cross"),      # r04 default
    ("latent_attnT8_kernel<true", "lat8_attn_self"), ("latent_attnT8_kernel<false", "lat8_attn_cross"),
    ("latent_attn_kernel<true", "lat_attn_self"), ("latent_attn_kernel<false", "lat_attn_cross"),
    ("latent_attn_fp8_kernel<true>", "lat8_attn_self"), ("latent_attn_fp8_kernel<false>", "lat8_attn_cross"),
    ("dec_qqt_kernel", "dec_qqt"), ("enc_attn_mfma_kernel", "enc_attn_mfma"), ("enc_attn2_kernel", "enc_attn_mfma"),
    ("layernorm_kernel", "layernorm"), ("ln_prep_kernel", "ln_prep"),
    ("dec_add_ln_kernel", "dec_add_ln"), ("dec_token_kernel", "dec_token"), ("gemm_wide2_kernel", "gemm_enc_layers(wide2)"),
    # the persistent encoder GEMM by epilogue: <1 = bias (QKV), <2 = bias + GELU (FC1), <3 = bias + fp32 residual (O-proj and FC2)
    ("gemm_pers_kernel<1", "gemm_enc_qkv"), ("gemm_pers_kernel<2", "gemm_enc_fc1"), ("gemm_pers_kernel<3", "gemm_enc_oproj+fc2"),
]


def klass(name):
    for sub, c in CLASSES:
        if sub in name:
            return c
    return None


def find(path, suffix):
    for d, _, files in os.walk(path):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(d, f)
    raise FileNotFoundError(f"no *{suffix} under {path}")


def pmc(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(find(path, "counter_collection.csv"))):
        c = klass(r["Kernel_Name"])
        if c:
            a = agg[c]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in agg.items()}


def algorithmic_bytes(c, rows, max_len=300):
    """SURVEY.md §8(d) / DESIGN.md §4: 1,536 B per key per crop + Qt in / Et out (12 heads x 768 x 2 B each way)."""
    if c == "lat_attn_cross":
        return rows * (197 * 1536 + 2 * 12 * 768 * 2)
    if c == "lat_attn_self":      # context grows 1..299: mean (max_len)/2 keys per launch
        return rows * (max_len / 2 * 1536 + 2 * 12 * 768 * 2)
    # encoder layer GEMMs (M = rows * 197): bf16 operands once, output once (+ the fp32 residual read for O-proj / FC2)
    M = rows * 197
    if c == "gemm_enc_qkv":
        return (M * 768 + 2304 * 768) * 2 + M * 2304 * 2
    if c == "gemm_enc_fc1":
        return (M * 768 + 3072 * 768) * 2 + M * 3072 * 2
    if c == "gemm_enc_oproj+fc2":  # equal launch counts: the mean of the two; + the bf16 copy of the rows (LayerNorm folding, r03)
        return ((M * 768 + 768 * 768) * 2 + (M * 3072 + 768 * 3072) * 2) / 2 + 2 * M * 768 * 4 + M * 768 * 2
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r02")
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq", help="directory of the SQ pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_LDS ...): <round>_mfma_util.json")
    ap.add_argument("--note", default="")
    ap.add_argument("--rows", type=int, required=True, help="rows of the internal batches the passes ran")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    classes = {}
    if a.stats:
        shutil.copy(find(a.stats, "kernel_stats.csv"), os.path.join(out, f"{a.round}_kernel_stats.csv"))
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(find(a.stats, "kernel_trace.csv"))):
            c = klass(r["Kernel_Name"])
            if c:
                agg[c].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for c, d in agg.items():
            classes[c] = {"calls": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d)}
            alg = algorithmic_bytes(c, a.rows)
            if alg:
                classes[c]["algorithmic_bytes_per_launch"] = alg
                classes[c]["achieved_GBps"] = alg / classes[c]["avg_ns"]
                classes[c]["frac_of_8TBps"] = alg / classes[c]["avg_ns"] / 8000.0
        json.dump({"note": a.note, "rows": a.rows, "classes": classes}, open(os.path.join(out, f"{a.round}_kernel_classes.json"), "w"), indent=1)
    if a.fetch and a.write:
        fe, wr = pmc(a.fetch), pmc(a.write)
        path = os.path.join(out, f"{a.round}_pmc_traffic.json")
        res = json.load(open(path)) if os.path.exists(path) else {}
        res.setdefault("correction", "traffic_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads half of a 16 B/lane stream)")
        res["note"] = a.note
        by = res.setdefault("by_rows", {}).setdefault(str(a.rows), {})
        for c, (n, v) in fe.items():
            w = wr.get(c, (0, 0.0))[1]
            ent = {"launches": n, "fetch_size_kb_avg": v, "write_size_kb_avg": w, "traffic_bytes_per_launch": (2 * v + w) * 1024}
            alg = algorithmic_bytes(c, a.rows)
            if alg:
                ent["algorithmic_bytes_per_launch"] = alg
                ent["traffic_over_algorithmic"] = ent["traffic_bytes_per_launch"] / alg
            by[c] = ent
        json.dump(res, open(path, "w"), indent=1)
    if a.sq:
        # one row per (dispatch, counter): per kernel class the mean of every counter.  MFMA utilisation = the matrix pipes' busy
        # cycles over the cycles they could have been busy: SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's SIMDs, in cycles)
        # / (4 SIMDs x CUs x the kernel's cycles), the kernel's cycles taken as GRBM-free: SQ_BUSY_CYCLES (quad-cycle units
        # of the busy SEs, MI355X_MICROARCH.md 'rocprofv3 PMC slots') - both ratios are reported raw next to the recipe.
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
        for r in csv.DictReader(open(find(a.sq, "counter_collection.csv"))):
            c = klass(r["Kernel_Name"])
            if c:
                e = agg[c][r["Counter_Name"]]
                e[0] += 1
                e[1] += float(r["Counter_Value"])
        res = {"note": a.note, "rows": a.rows,
               "recipe": "rocprofv3 --pmc <SQ counters> (a pass of its own, no trace) of `python bench.py --steps 10 --warmup 0 --lanes 1 --only-timed`; "
                         "per kernel class the mean per dispatch; mfma_util_of_peak_clock = SQ_VALU_MFMA_BUSY_CYCLES / (256 CUs x 4 SIMDs) / (launch duration x 2.4 GHz), "
                         "the launch duration from the kernel trace of the same round (a bf16 16x16x32 MFMA keeps its SIMD's matrix pipe busy for 16 cycles: "
                         "the ratio is the fraction of the 2.5 PFLOP/s dense peak the launch's MFMA instructions account for); "
                         "mfma_busy_over_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES as the counters come",
               "classes": {}}
        for c, d in agg.items():
            ent = {k: v[1] / v[0] for k, v in d.items()}
            ent["dispatches"] = max(v[0] for v in d.values())
            if ent.get("SQ_BUSY_CYCLES"):
                ent["mfma_busy_over_busy"] = ent.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / ent["SQ_BUSY_CYCLES"]
            # matrix-pipe utilisation proper: busy cycles per SIMD (the counter sums the chip's 256 CUs x 4 SIMDs) over the cycles
            # of the launch at the 2.4 GHz the 2.5 PF peak is priced at - duration from THIS round's kernel trace (--stats)
            if c in classes and ent.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                ent["avg_ns_from_trace"] = classes[c]["avg_ns"]
                ent["mfma_util_of_peak_clock"] = ent["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (classes[c]["avg_ns"] * 2.4)
            res["classes"][c] = ent
        json.dump(res, open(os.path.join(out, f"{a.round}_mfma_util.json"), "w"), indent=1)
    print(sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
