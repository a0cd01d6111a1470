#!/usr/bin/env python3
"""Build libmocr_hip.so (HIP kernels + C ABI) for gfx950, in-tree.

    python manga-ocr_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so lands in manga_ocr/_lib/ (git-ignored, but it
travels to the GPU box with the repo snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "manga_ocr", "_lib")
OUT = os.path.join(OUT_DIR, "libmocr_hip.so")
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


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-o", OUT] + SOURCES
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
