#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/<dir>/*.csv) into the small files kept under profiles/.

    python tools/summarize_profiles.py --round r01 --stats gpurun_out/prof3 --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write

* <round>_kernel_stats.csv : rocprofv3 --kernel-trace --stats summary, verbatim (per-kernel calls / avg ns)
* <round>_pmc_traffic.json : per kernel, average FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes)
  and the corrected HBM traffic:  (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes - on gfx950 FETCH_SIZE counts
  half of a wide (16 B/lane) coalesced read stream (MI355X_MICROARCH.md, HBM section).
"""
import argparse
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc(path):
    f = [p for p in os.listdir(path) if p.endswith("counter_collection.csv")][0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(os.path.join(path, f))):
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r01")
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--note", default="")
    ap.add_argument("--rows", type=int, default=0, help="rows of the merged batch the PMC passes ran")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    if a.stats:
        f = [p for p in os.listdir(a.stats) if p.endswith("kernel_stats.csv")][0]
        shutil.copy(os.path.join(a.stats, f), os.path.join(out, f"{a.round}_kernel_stats.csv"))
    if a.fetch and a.write:
        fe, wr = pmc(a.fetch), pmc(a.write)
        res = {"note": a.note, "rows": a.rows, "correction": "traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads half of a 16 B/lane stream)",
               "kernels": {}}
        for k, (n, v) in sorted(fe.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            w = wr.get(k, (0, 0.0))[1]
            res["kernels"][k] = {"launches": n, "fetch_size_kb_avg": v, "write_size_kb_avg": w,
                                 "traffic_bytes": (2 * v + w) * 1024}
        json.dump(res, open(os.path.join(out, f"{a.round}_pmc_traffic.json"), "w"), indent=1)
    print(os.listdir(out))


if __name__ == "__main__":
    main()
