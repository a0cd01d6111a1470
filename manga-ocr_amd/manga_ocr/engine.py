"""Python handle on the native engine (libmocr_hip.so) - thin, typed wrappers over the C ABI."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _capi
from .weights import DEFAULT_SPEC, ModelSpec, check_weights

DTYPES = {"fp32": _capi.MOCR_F32, "f32": _capi.MOCR_F32, "float32": _capi.MOCR_F32,
          "bf16": _capi.MOCR_BF16, "bfloat16": _capi.MOCR_BF16}


def _ptr(a) -> C.c_void_p:
    """numpy array (host) / torch tensor (host or device) / int address -> void*"""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    raise TypeError(f"cannot take the address of {type(a)}")


def device_memory(device: int = 0) -> Tuple[int, int]:
    """(free, total) bytes of HBM on a device."""
    lib = _capi.load_library()
    f, t = C.c_int64(0), C.c_int64(0)
    rc = lib.mocr_device_memory(device, C.byref(f), C.byref(t))
    if rc != _capi.MOCR_OK:
        raise _capi.MocrError(f"mocr_device_memory({device}) failed with code {rc} (is a MI355X visible to this process?)")
    return int(f.value), int(t.value)


class Engine:
    """One engine = one GPU = one HIP stream.  Thread-safe (calls are serialised natively)."""

    def __init__(self, weights: Dict[str, np.ndarray], spec: ModelSpec = DEFAULT_SPEC, *, dtype: str = "bf16",
                 device: int = 0, max_batch: int = 64, flags: int = 0, lanes: int = 1,
                 lib_path: Optional[str] = None):
        self.lib = _capi.load_library(lib_path)
        self.spec = spec
        self.dtype = dtype
        self.max_batch = int(max_batch)
        self.device = int(device)
        check_weights(weights, spec)
        cfg = _capi.MocrConfig(
            struct_size=C.sizeof(_capi.MocrConfig), device=device, dtype=DTYPES[dtype], max_batch=max_batch,
            max_len=spec.max_len, image_size=spec.image_size, patch_size=spec.patch_size, hidden=spec.hidden,
            enc_layers=spec.enc_layers, dec_layers=spec.dec_layers, heads=spec.heads, ffn=spec.ffn, vocab=spec.vocab,
            max_pos=spec.max_pos, start_id=spec.start_id, eos_id=spec.eos_id, pad_id=spec.pad_id,
            ln_eps=spec.ln_eps, flags=flags, lanes=lanes)
        h = C.c_void_p()
        rc = self.lib.mocr_create(C.byref(cfg), C.byref(h))
        if rc != _capi.MOCR_OK or not h.value:
            raise _capi.MocrError(f"mocr_create failed with code {rc} (is a MI355X visible to this process?)")
        self._h = h
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(self.lib.mocr_set_tensor(self._h, name.encode(), _ptr(a), shape, a.ndim))
        self._check(self.lib.mocr_commit_weights(self._h))

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int) -> None:
        if rc != _capi.MOCR_OK:
            msg = self.lib.mocr_last_error(self._h)
            raise _capi.MocrError(f"libmocr_hip error {rc}: {msg.decode(errors='replace') if msg else ''}")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.mocr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return int(self.lib.mocr_stream(self._h) or 0)

    def synchronize(self) -> None:
        self._check(self.lib.mocr_synchronize(self._h))

    # ------------------------------------------------------------------ the hot path
    def recognize(self, images: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """images uint8 [n,224,224] (luminance) or [n,224,224,3] (RGB), host memory.
        Returns (ids int32 [n,max_len], lengths int32 [n])."""
        a = np.ascontiguousarray(images, dtype=np.uint8)
        if a.ndim == 3:
            n, h, w = a.shape
            ch = 1
        elif a.ndim == 4 and a.shape[3] == 3:
            n, h, w, ch = a.shape
        else:
            raise ValueError("images must be uint8 [n,h,w] or [n,h,w,3]")
        ids = np.zeros((n, self.spec.max_len), dtype=np.int32)
        lens = np.zeros(n, dtype=np.int32)
        self._check(self.lib.mocr_recognize(self._h, _ptr(a), n, h, w, w * ch, h * w * ch, ch, _ptr(ids), _ptr(lens)))
        return ids, lens

    def _image_descs(self, images, bgr: bool = False, rotate=None):
        """numpy uint8 [h,w] (L) or [h,w,3] (RGB; B,G,R order when `bgr`) arrays of any sizes -> (ctypes array of
        mocr_image, keep-alive list).  Arrays with contiguous pixels and any row stride are passed as they are
        (a crop that is a view into a page is not copied here).  ``rotate``: per image 0 / ROTATE_90_CW / ROTATE_90_CCW -
        the reference's orientation-only rotation, done by the device's resize addressing."""
        keep = []
        descs = (_capi.MocrImage * len(images))()
        for i, im in enumerate(images):
            a = im if isinstance(im, np.ndarray) and im.dtype == np.uint8 else np.ascontiguousarray(im, dtype=np.uint8)
            if a.ndim == 2:
                ch = 1
            elif a.ndim == 3 and a.shape[2] == 3:
                ch = 3
            else:
                raise ValueError("each image must be uint8 [h,w] or [h,w,3]")
            pix_ok = a.strides[1] == ch and (ch == 1 or a.strides[2] == 1) and a.strides[0] >= a.shape[1] * ch
            if not pix_ok:
                a = np.ascontiguousarray(a)
            keep.append(a)
            d = descs[i]
            d.data = a.__array_interface__["data"][0]        # (a.ctypes.data builds a ctypes object per call: 10x slower)
            d.height, d.width = a.shape[0], a.shape[1]
            d.row_stride = a.strides[0]
            d.channels = _capi.CHANNELS_BGR if (bgr and ch == 3) else ch
            d.rotate = int(rotate[i]) if rotate is not None else 0
        return descs, keep

    def recognize_images(self, images, bgr: bool = False, rotate=None) -> Tuple[np.ndarray, np.ndarray]:
        """Crops of any sizes (list of uint8 [h,w] / [h,w,3] arrays; `bgr`: 3-channel crops are in OpenCV order;
        `rotate`: per crop 0 / 1 (90 degrees clockwise) / 2 (counter-clockwise), applied on the device): luminance
        conversion and the Pillow-exact BILINEAR resize to 224x224 run on the device.
        Returns (ids int32 [n,max_len], lengths int32 [n])."""
        if len(images) == 0:
            return np.zeros((0, self.spec.max_len), dtype=np.int32), np.zeros(0, dtype=np.int32)
        descs, keep = self._image_descs(images, bgr, rotate)
        n = len(keep)
        ids = np.zeros((n, self.spec.max_len), dtype=np.int32)
        lens = np.zeros(n, dtype=np.int32)
        self._check(self.lib.mocr_recognize_images(self._h, descs, n, _ptr(ids), _ptr(lens)))
        return ids, lens

    def recognize_regions(self, pages, regions, bgr: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """pages: list of uint8 [H,W,3] (or [H,W]) arrays; regions: iterable of (page_index, x, y, w, h) bounding
        rectangles.  Each page is uploaded once; the 8 %-padded, page-clipped crop of every region
        (``src/ui/main_window.py:9530-9540``) is cut on the device.  Returns (ids [n,max_len], lengths [n]);
        a region reduced to a sliver has length 0."""
        regs = list(regions)
        n = len(regs)
        ids = np.zeros((n, self.spec.max_len), dtype=np.int32)
        lens = np.zeros(n, dtype=np.int32)
        if n == 0:
            return ids, lens
        descs, keep = self._image_descs(pages, bgr)
        arr = (_capi.MocrRegion * n)()
        for i, (pg, x, y, w, h) in enumerate(regs):
            arr[i].page, arr[i].x, arr[i].y, arr[i].width, arr[i].height = int(pg), int(x), int(y), int(w), int(h)
        self._check(self.lib.mocr_recognize_regions(self._h, descs, len(keep), arr, n, _ptr(ids), _ptr(lens)))
        return ids, lens

    def graph_count(self) -> int:
        return int(self.lib.mocr_graph_count(self._h))

    def ln_fold_state(self) -> Tuple[bool, float]:
        """(folds, noise_ratio): whether fat bf16 batches run with the encoder's LayerNorms folded into the GEMMs, and the
        input-rounding noise ratio commit measured on this checkpoint's residual stream (include/mocr.h)."""
        r = C.c_float(0.0)
        on = self.lib.mocr_ln_fold_state(self._h, C.byref(r))
        return bool(on), float(r.value)

    def decode_slot_steps(self) -> int:
        """Decode slots x steps enqueued so far (what the decode launches were sized for, in row-steps)."""
        return int(self.lib.mocr_decode_slot_steps(self._h))

    def compaction_count(self) -> int:
        """Row compactions performed so far: unfinished rows moved to the first decode slots between chunks of steps."""
        return int(self.lib.mocr_compaction_count(self._h))

    def preprocess(self, images, bgr: bool = False, rotate=None) -> np.ndarray:
        """Test hook: the uint8 [n,224,224] planes the encoder sees for these crops."""
        descs, keep = self._image_descs(images, bgr, rotate)
        out = np.zeros((len(keep), self.spec.image_size, self.spec.image_size), dtype=np.uint8)
        self._check(self.lib.mocr_preprocess(self._h, descs, len(keep), _ptr(out)))
        return out

    def recognize_device(self, d_gray, n: int, d_out_ids, d_out_len) -> None:
        """Asynchronous; all three are device buffers (torch CUDA tensors or raw addresses)."""
        self._check(self.lib.mocr_recognize_device(self._h, _ptr(d_gray), n, _ptr(d_out_ids), _ptr(d_out_len)))

    def set_generate_max_length(self, max_len: int) -> None:
        self._check(self.lib.mocr_set_generate_max_length(self._h, int(max_len)))

    def recognize_gray(self, gray: np.ndarray, max_len: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        a = np.ascontiguousarray(gray, dtype=np.uint8)
        n = a.shape[0]
        ids = np.zeros((n, self.spec.max_len), dtype=np.int32)
        lens = np.zeros(n, dtype=np.int32)
        self._check(self.lib.mocr_recognize_gray_host(self._h, _ptr(a), n, max_len or self.spec.max_len, _ptr(ids), _ptr(lens)))
        return ids, lens

    # ------------------------------------------------------------------ test hooks
    def encode(self, d_gray, n: int) -> np.ndarray:
        out = np.zeros((n, self.spec.enc_tokens, self.spec.hidden), dtype=np.float32)
        self._check(self.lib.mocr_encode(self._h, _ptr(d_gray), n, _ptr(out)))
        return out

    def decode_logits(self, d_gray, n: int, forced_ids: np.ndarray) -> np.ndarray:
        f = np.ascontiguousarray(forced_ids, dtype=np.int32)
        T = f.shape[1]
        out = np.zeros((n, T, self.spec.vocab), dtype=np.float32)
        self._check(self.lib.mocr_decode_logits(self._h, _ptr(d_gray), n, _ptr(f), T, _ptr(out)))
        return out

    def op_gemm(self, dA, dW, d_bias, d_out, d_resid, M, N, K, epilogue, tile=128, split_k=1) -> None:
        self._check(self.lib.mocr_op_gemm(self._h, _ptr(dA), _ptr(dW), _ptr(d_bias), _ptr(d_out), _ptr(d_resid),
                                          M, N, K, epilogue, tile, split_k))

    def op_gemm_ln(self, dA, dW, d_bias, d_out, d_resid, M, N, K, epilogue, tile, d_part, d_csum, d_xb) -> None:
        """The persistent encoder GEMM with the LayerNorm folded in (include/mocr.h, mocr_op_gemm_ln)."""
        self._check(self.lib.mocr_op_gemm_ln(self._h, _ptr(dA), _ptr(dW), _ptr(d_bias), _ptr(d_out), _ptr(d_resid), M, N, K, epilogue, tile,
                                             _ptr(d_part), _ptr(d_csum), _ptr(d_xb)))

    def op_ln_prep(self, d_x, d_xb, d_part, M) -> None:
        self._check(self.lib.mocr_op_ln_prep(self._h, _ptr(d_x), _ptr(d_xb), _ptr(d_part), M))

    def op_layernorm(self, d_x, d_gamma, d_beta, d_out, M) -> None:
        self._check(self.lib.mocr_op_layernorm(self._h, _ptr(d_x), _ptr(d_gamma), _ptr(d_beta), _ptr(d_out), M))

    def op_enc_attention(self, d_qkv, d_ctx, n, impl) -> None:
        self._check(self.lib.mocr_op_enc_attention(self._h, _ptr(d_qkv), _ptr(d_ctx), n, impl))

    def op_latent_attention(self, d_qt, d_x, d_out, n, length, x_batch_stride) -> None:
        self._check(self.lib.mocr_op_latent_attention(self._h, _ptr(d_qt), _ptr(d_x), _ptr(d_out), n, length, x_batch_stride))

    def op_quant_fp8(self, d_x, d_x8, n_elems: int, inv_sx: float) -> None:
        self._check(self.lib.mocr_op_quant_fp8(self._h, _ptr(d_x), _ptr(d_x8), n_elems, inv_sx))

    def op_latent_attention_fp8(self, d_qt, d_x8, d_out, n, length, x_batch_stride_bytes, sx) -> None:
        self._check(self.lib.mocr_op_latent_attention_fp8(self._h, _ptr(d_qt), _ptr(d_x8), _ptr(d_out), n, length, x_batch_stride_bytes, sx))

    # ------------------------------------------------------------------ per-kernel timing
    def op_qqt(self, d_x, d_wq, d_bq, d_wkT, d_qt, n: int) -> None:
        self._check(self.lib.mocr_op_qqt(self._h, _ptr(d_x), _ptr(d_wq), _ptr(d_bq), _ptr(d_wkT), _ptr(d_qt), n))

    def profile_enable(self, on: bool = True) -> None:
        self._check(self.lib.mocr_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self) -> None:
        self._check(self.lib.mocr_profile_reset(self._h))

    def profile_get(self) -> List[dict]:
        cap = 64
        arr = (_capi.MocrKernelStat * cap)()
        n = C.c_int32(0)
        self._check(self.lib.mocr_profile_get(self._h, arr, cap, C.byref(n)))
        return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches), total_ms=float(arr[i].total_ms),
                     flops=float(arr[i].flops), bytes=float(arr[i].bytes)) for i in range(n.value)]
