"""ctypes binding of include/mocr.h (libmocr_hip.so).  No fallback: if the library or a GPU is
missing, importing works but constructing an engine raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libmocr_hip.so")

MOCR_OK = 0
MOCR_F32, MOCR_BF16 = 0, 1
FLAG_SIMPLE_ATTENTION, FLAG_NO_GRAPH, FLAG_NO_EARLY_EXIT, FLAG_CLASSIC_ATTENTION = 1, 2, 4, 8
FLAG_NO_FUSED_ARGMAX, FLAG_NO_FUSED_QQT, FLAG_LATENT_ALWAYS, FLAG_FP8_ATTENTION = 16, 32, 64, 128
FLAG_NO_SMALL_BATCH_PATH = 256
FLAG_NO_LN_FOLD = 512
EPI_SLAB, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_PATCH, EPI_BIAS_F32 = range(6)


class MocrConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "device", "dtype", "max_batch", "max_len", "image_size", "patch_size", "hidden",
        "enc_layers", "dec_layers", "heads", "ffn", "vocab", "max_pos", "start_id", "eos_id", "pad_id")] + \
        [("ln_eps", C.c_float), ("flags", C.c_int32), ("lanes", C.c_int32)]


class MocrKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


# every symbol include/mocr.h declares: name -> (restype, argtypes)
_P = C.c_void_p
class MocrImage(C.Structure):
    """include/mocr.h: mocr_image"""
    _fields_ = [("data", C.c_void_p), ("height", C.c_int32), ("width", C.c_int32), ("row_stride", C.c_int64),
                ("channels", C.c_int32), ("rotate", C.c_int32)]


class MocrRegion(C.Structure):
    """include/mocr.h: mocr_region"""
    _fields_ = [(n, C.c_int32) for n in ("page", "x", "y", "width", "height")]


CHANNELS_BGR = -3
ROTATE_NONE, ROTATE_90_CW, ROTATE_90_CCW = 0, 1, 2

SYMBOLS = {
    "mocr_abi_version": (C.c_int, []),
    "mocr_create": (C.c_int, [C.POINTER(MocrConfig), C.POINTER(_P)]),
    "mocr_destroy": (None, [_P]),
    "mocr_last_error": (C.c_char_p, [_P]),
    "mocr_set_tensor": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32]),
    "mocr_commit_weights": (C.c_int, [_P]),
    "mocr_recognize": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, _P, _P]),
    "mocr_recognize_images": (C.c_int, [_P, C.POINTER(MocrImage), C.c_int32, _P, _P]),
    "mocr_recognize_regions": (C.c_int, [_P, C.POINTER(MocrImage), C.c_int32, C.POINTER(MocrRegion), C.c_int32, _P, _P]),
    "mocr_graph_count": (C.c_int, [_P]),
    "mocr_compaction_count": (C.c_int64, [_P]),
    "mocr_decode_slot_steps": (C.c_int64, [_P]),
    "mocr_ln_fold_state": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "mocr_device_memory": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mocr_preprocess": (C.c_int, [_P, C.POINTER(MocrImage), C.c_int32, _P]),
    "mocr_recognize_device": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "mocr_set_generate_max_length": (C.c_int, [_P, C.c_int32]),
    "mocr_synchronize": (C.c_int, [_P]),
    "mocr_stream": (_P, [_P]),
    "mocr_encode": (C.c_int, [_P, _P, C.c_int32, _P]),
    "mocr_decode_logits": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P]),
    "mocr_recognize_gray_host": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    "mocr_op_gemm": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mocr_op_layernorm": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32]),
    "mocr_op_gemm_ln": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "mocr_op_ln_prep": (C.c_int, [_P, _P, _P, _P, C.c_int32]),
    "mocr_op_enc_attention": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32]),
    "mocr_op_latent_attention": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int64]),
    "mocr_op_quant_fp8": (C.c_int, [_P, _P, _P, C.c_int64, C.c_float]),
    "mocr_op_latent_attention_fp8": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int64, C.c_float]),
    "mocr_op_qqt": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32]),
    "mocr_profile_enable": (C.c_int, [_P, C.c_int32]),
    "mocr_profile_reset": (C.c_int, [_P]),
    "mocr_profile_get": (C.c_int, [_P, C.POINTER(MocrKernelStat), C.c_int32, C.POINTER(C.c_int32)]),
}

_lib: Optional[C.CDLL] = None


class MocrError(RuntimeError):
    pass


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libmocr_hip.so and bind every declared symbol (raises if one is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MOCR_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise MocrError(f"{p} not found: build it with `python manga-ocr_amd/build.py` "
                        "(the Manga-OCR engine has no CPU fallback)")
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.mocr_abi_version() != 2:
        raise MocrError("libmocr_hip.so ABI version mismatch")
    if path is None:
        _lib = lib
    return lib
