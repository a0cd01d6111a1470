"""Integer restatement (numpy) of the two Pillow operations in front of the recogniser.
TEST INFRASTRUCTURE ONLY (see oracle/mocr_oracle.py header for who may import oracle/).

The reference hands ``MangaOcr.__call__`` an RGB PIL image (``src/ui/main_window.py:9800``);
the recogniser then does ``img.convert('L').convert('RGB')`` and the HF image processor
resizes to 224x224 with PIL BILINEAR (TF/models/vit/image_processing_pil_vit.py:20-27,
TF/image_processing_backends.py:521-570).  Both are deterministic 8-bit fixed-point
routines of Pillow (third-party C code, Pillow 12.2.0 in the build container):

  convert('L')      libImaging/Convert.c   L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16
  resize(BILINEAR)  libImaging/Resample.c  separable triangle filter whose support scales
                    with the down-scale factor (antialiasing), coefficients rounded to
                    22 fractional bits, horizontal pass then vertical pass, each pass
                    rounded and clamped to uint8.

Pinned by tests/test_oracle_golden.py against Pillow itself (run in the build container
and whenever Pillow is importable) and against tests/golden/preprocess.npz.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def rgb_to_l(rgb: np.ndarray) -> np.ndarray:
    """uint8 [...,3] RGB -> uint8 [...] luminance (ITU-R 601-2, Pillow fixed point)."""
    r = rgb[..., 0].astype(np.uint32)
    g = rgb[..., 1].astype(np.uint32)
    b = rgb[..., 2].astype(np.uint32)
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def bilinear_coeffs(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc for the triangle filter (support 1.0).
    Returns (bounds int32 [out,2] = (xmin, count), kk int32 [out, ksize])."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C (int) cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        cnt = xmax - xmin
        w = np.zeros(cnt, dtype=np.float64)
        for x in range(cnt):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = w.sum() if cnt else 0.0
        # Pillow accumulates ww in a running double sum, in index order
        acc = 0.0
        for x in range(cnt):
            acc += w[x]
        ww = acc
        if ww != 0.0:
            w = w / ww
        for x in range(cnt):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, cnt)
    return bounds, kk


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    in_size = img.shape[axis]
    bounds, kk = bilinear_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, cnt = int(bounds[xx, 0]), int(bounds[xx, 1])
        k = kk[xx, :cnt].astype(np.int64)
        acc = np.tensordot(k, src[xmin:xmin + cnt], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img: np.ndarray, out_h: int = 224, out_w: int = 224) -> np.ndarray:
    """uint8 [h,w] -> uint8 [out_h,out_w], bit-exact with PIL.Image.resize((w,h), BILINEAR).
    ImagingResample runs the horizontal pass first, then the vertical pass, and skips a
    pass whose size is unchanged."""
    h, w = img.shape[:2]
    out = img
    if w != out_w:
        out = _resample_axis(out, out_w, axis=1)
    if h != out_h:
        out = _resample_axis(out, out_h, axis=0)
    return np.ascontiguousarray(out)


def preprocess_rgb_to_gray224(rgb: np.ndarray) -> np.ndarray:
    """uint8 [h,w,3] RGB -> uint8 [224,224]: what the encoder sees in each of its three
    (identical) input channels before the /255 and (x-0.5)/0.5 scaling."""
    return resize_bilinear_u8(rgb_to_l(rgb), 224, 224)
