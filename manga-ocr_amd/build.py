#!/usr/bin/env python3
"""Build libmocr_hip.so (HIP kernels + C ABI) for gfx950, in-tree.

    python manga-ocr_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so lands in manga_ocr/_lib/ (git-ignored, but it
travels to the GPU box with the repo snapshot).

    python manga-ocr_amd/build.py --experiments

builds manga_ocr/_lib/libmocr_hip_lab.so instead: the same engine plus the A/B kernels of earlier rounds and the MOCR_*
environment knobs (tools/ select it with MOCR_LIB=manga-ocr_amd/manga_ocr/_lib/libmocr_hip_lab.so)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "manga_ocr", "_lib")
OUT = os.path.join(OUT_DIR, "libmocr_hip.so")
OUT_LAB = os.path.join(OUT_DIR, "libmocr_hip_lab.so")     # --experiments: never the library the package loads by default
SOURCES = [os.path.join(CSRC, "engine.hip")]


def _deps():
    deps = list(SOURCES) + [os.path.join(HERE, "..", "include", "mocr.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return deps


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False, verbose: bool = True, experiments: bool = False) -> str:
    """experiments: -DMOCR_EXPERIMENTS - the A/B kernels of earlier rounds (csrc/kernels_gemm_lab.h) and the MOCR_*
    environment knobs, for tools/; the product library has neither."""
    if not force and not experiments and not needs_build():
        return OUT
    out = OUT_LAB if experiments else OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-o", out] + (["-DMOCR_EXPERIMENTS"] if experiments else []) + SOURCES
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experiments="--experiments" in sys.argv))
