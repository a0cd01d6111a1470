"""Model specification and weight sources for the Manga-OCR recogniser.

The recogniser the reference calls at ``src/ui/main_window.py:9801`` is a
VisionEncoderDecoder model: ViT-B/16-224 encoder + 2-layer BERT decoder with
cross-attention.  This module owns

* ``ModelSpec``       - the hyper-parameters (fixed by BASELINE.json),
* ``tensor_table()``  - every parameter tensor the engine consumes, by its
                        canonical (transformers-5.x state_dict) name,
* ``synthetic_weights()`` - the deterministic synthetic weights used by every
                        pinned test and by bench.py (no checkpoint is available
                        offline),
* ``load_checkpoint()``   - loader for a local HF ``model_dir`` (4.x or 5.x key
                        spelling, ``TF/conversion_mapping.py:338-346``).

Weights are plain ``dict[str, np.ndarray(float32)]``; nothing here touches a GPU.
"""
from __future__ import annotations

import json
import os
import re
from dataclasses import dataclass, field, replace, asdict
from typing import Dict, List, Tuple

import numpy as np


@dataclass(frozen=True)
class ModelSpec:
    image_size: int = 224
    patch_size: int = 16
    hidden: int = 768          # encoder == decoder width, so no enc_to_dec_proj
    enc_layers: int = 12
    dec_layers: int = 2
    heads: int = 12
    ffn: int = 3072
    vocab: int = 6144
    max_pos: int = 512         # decoder position table
    type_vocab: int = 2
    ln_eps: float = 1e-12
    max_len: int = 300         # generate(max_length=300)
    start_id: int = 2          # [CLS] = decoder_start_token_id
    eos_id: int = 3            # [SEP]
    pad_id: int = 0            # [PAD]
    # generation settings a checkpoint's config.json asked for and this engine does NOT honour (it decodes greedily, as
    # BASELINE.json's north_star fixes): e.g. (("num_beams", 4), ("no_repeat_ngram_size", 3)).  Readable by callers via
    # MangaOcr.ignored_generation_config; empty for synthetic weights and for greedy checkpoints.
    ignored_generation: tuple = field(default=(), compare=False)

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def n_patches(self) -> int:
        return self.grid * self.grid

    @property
    def enc_tokens(self) -> int:
        return self.n_patches + 1

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    def to_dict(self) -> dict:
        return asdict(self)


DEFAULT_SPEC = ModelSpec()

# kind: 'w' dense weight / embedding table, 'b' bias, 'g' LayerNorm gain, 'beta' LayerNorm shift
TensorRow = Tuple[str, Tuple[int, ...], str]


def tensor_table(spec: ModelSpec = DEFAULT_SPEC) -> List[TensorRow]:
    """Ordered list of (canonical name, shape, kind).  The ORDER is part of the
    synthetic-weight definition: tensors are drawn from one RandomState stream in
    exactly this order."""
    D, F, V = spec.hidden, spec.ffn, spec.vocab
    P = spec.patch_size
    rows: List[TensorRow] = []
    add = rows.append
    e = "encoder."
    add((e + "embeddings.cls_token", (1, 1, D), "w"))
    add((e + "embeddings.position_embeddings", (1, spec.enc_tokens, D), "w"))
    add((e + "embeddings.patch_embeddings.projection.weight", (D, 3, P, P), "w"))
    add((e + "embeddings.patch_embeddings.projection.bias", (D,), "b"))
    for i in range(spec.enc_layers):
        p = f"{e}layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            add((p + f"attention.{n}.weight", (D, D), "w"))
            add((p + f"attention.{n}.bias", (D,), "b"))
        add((p + "layernorm_before.weight", (D,), "g"))
        add((p + "layernorm_before.bias", (D,), "beta"))
        add((p + "layernorm_after.weight", (D,), "g"))
        add((p + "layernorm_after.bias", (D,), "beta"))
        add((p + "mlp.fc1.weight", (F, D), "w"))
        add((p + "mlp.fc1.bias", (F,), "b"))
        add((p + "mlp.fc2.weight", (D, F), "w"))
        add((p + "mlp.fc2.bias", (D,), "b"))
    add((e + "layernorm.weight", (D,), "g"))
    add((e + "layernorm.bias", (D,), "beta"))
    d = "decoder.bert."
    add((d + "embeddings.word_embeddings.weight", (V, D), "w"))
    add((d + "embeddings.position_embeddings.weight", (spec.max_pos, D), "w"))
    add((d + "embeddings.token_type_embeddings.weight", (spec.type_vocab, D), "w"))
    add((d + "embeddings.LayerNorm.weight", (D,), "g"))
    add((d + "embeddings.LayerNorm.bias", (D,), "beta"))
    for i in range(spec.dec_layers):
        p = f"{d}encoder.layer.{i}."
        for blk in ("attention", "crossattention"):
            for n in ("query", "key", "value"):
                add((p + f"{blk}.self.{n}.weight", (D, D), "w"))
                add((p + f"{blk}.self.{n}.bias", (D,), "b"))
            add((p + f"{blk}.output.dense.weight", (D, D), "w"))
            add((p + f"{blk}.output.dense.bias", (D,), "b"))
            add((p + f"{blk}.output.LayerNorm.weight", (D,), "g"))
            add((p + f"{blk}.output.LayerNorm.bias", (D,), "beta"))
        add((p + "intermediate.dense.weight", (F, D), "w"))
        add((p + "intermediate.dense.bias", (F,), "b"))
        add((p + "output.dense.weight", (D, F), "w"))
        add((p + "output.dense.bias", (D,), "b"))
        add((p + "output.LayerNorm.weight", (D,), "g"))
        add((p + "output.LayerNorm.bias", (D,), "beta"))
    c = "decoder.cls.predictions."
    add((c + "transform.dense.weight", (D, D), "w"))
    add((c + "transform.dense.bias", (D,), "b"))
    add((c + "transform.LayerNorm.weight", (D,), "g"))
    add((c + "transform.LayerNorm.bias", (D,), "beta"))
    # The vocabulary projection.  HF ties it to the word embeddings when
    # tie_word_embeddings is set (TF/models/bert/modeling_bert.py:825-828); the engine
    # always receives it as its own tensor.
    add((c + "decoder.weight", (V, D), "w"))
    add((c + "decoder.bias", (V,), "b"))
    return rows


def synthetic_weights(seed: int = 0, spec: ModelSpec = DEFAULT_SPEC, *, std: float = 0.02,
                      bias_std: float = 0.02, ln_std: float = 0.1, tie_lm_head: bool = True,
                      eos_bias: float = 0.0, logit_scale: float = 1.0, vocab_bias_std: float = 0.0,
                      hostile=False) -> Dict[str, np.ndarray]:
    """Deterministic synthetic parameters (float32).

    One ``numpy.random.RandomState(seed)`` stream (frozen algorithm), tensors drawn
    in ``tensor_table`` order with ``standard_normal(shape)``:
      'w'    -> std * z            'b'    -> bias_std * z
      'g'    -> 1 + ln_std * z     'beta' -> ln_std * z
    Biases and LayerNorm parameters are deliberately non-trivial so that a kernel
    which drops one fails parity.  ``tie_lm_head`` copies the word-embedding table
    into the vocabulary projection after drawing (the draw still happens, so the
    stream position of later tensors does not depend on the flag).
    ``eos_bias`` is added to the vocabulary bias of ``eos_id`` (makes rows finish at
    different steps: exercises the finished-row padding rule).  ``logit_scale``
    multiplies the vocabulary projection weight (it scales the logits' rounding noise
    by the same factor, so it does NOT make a bf16 engine's ids more comparable with an
    fp32 oracle's - kept for completeness).  ``vocab_bias_std`` adds an independent
    N(0, vocab_bias_std^2) term (its own RandomState(seed + 7919) stream) to the fp32
    vocabulary bias: exact in every engine, it widens the top-2 margins relative to the
    bf16 noise of the matrix products - the widened-margin weights of the bf16 id tests.
    ``hostile`` (r04) gives the ENCODER the residual-stream statistics of trained ViTs instead of the near-normalised stream
    plain N(0, std^2) weights produce (its own RandomState(seed + 104729) stream, applied after the draws above):
    LayerNorm gains ~ U(0.2, 3) and shifts ~ N(0, 0.5^2); a DC offset of +2.0 (about 6 sigma of the embedded patches) on
    every channel of the patch-embedding bias and the CLS token; two massive-activation channels (HOSTILE_CHANNELS) at
    +100 (about 300 sigma).  Nothing in a pre-LN ViT ever removes what the embeddings put on the stream, so all of it
    reaches every LayerNorm - the inputs on which shortcuts that round the raw stream to bf16 are to be judged.
    ``hostile="dc"``: the same gains / shifts and a DC offset of +4.0 WITHOUT the massive channels - the row mean is then
    several times the row's spread at every LayerNorm (with massive channels the spread is theirs), the case in which
    rounding the raw stream costs precision where rounding the normalised stream does not.
    """
    rs = np.random.RandomState(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape, kind in tensor_table(spec):
        z = rs.standard_normal(shape)
        if kind == "w":
            a = std * z
        elif kind == "b":
            a = bias_std * z
        elif kind == "g":
            a = 1.0 + ln_std * z
        elif kind == "beta":
            a = ln_std * z
        else:  # pragma: no cover
            raise ValueError(kind)
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    lm_w = "decoder.cls.predictions.decoder.weight"
    lm_b = "decoder.cls.predictions.decoder.bias"
    if tie_lm_head:
        out[lm_w] = out["decoder.bert.embeddings.word_embeddings.weight"].copy()
    if logit_scale != 1.0:
        out[lm_w] = (out[lm_w] * np.float32(logit_scale)).astype(np.float32)
    if vocab_bias_std != 0.0:
        extra = np.random.RandomState(seed + 7919).standard_normal(spec.vocab) * vocab_bias_std
        extra[[spec.pad_id, spec.eos_id]] = -abs(extra).max()      # rows keep decoding: neither EOS nor PAD gets a head start
        out[lm_b] = (out[lm_b] + extra.astype(np.float32)).astype(np.float32)
    if eos_bias != 0.0:
        out[lm_b] = out[lm_b].copy()
        out[lm_b][spec.eos_id] += np.float32(eos_bias)
    if hostile:
        hs = np.random.RandomState(seed + 104729)
        for name, shape, kind in tensor_table(spec):
            if not name.startswith("encoder."):
                continue
            if kind == "g":
                out[name] = hs.uniform(0.2, 3.0, size=shape).astype(np.float32)
            elif kind == "beta":
                out[name] = (0.5 * hs.standard_normal(shape)).astype(np.float32)
        pb = "encoder.embeddings.patch_embeddings.projection.bias"
        cls = "encoder.embeddings.cls_token"
        dc_only = hostile == "dc"
        out[pb] = out[pb] + np.float32(HOSTILE_DC_ONLY if dc_only else HOSTILE_DC)
        out[cls] = out[cls] + np.float32(HOSTILE_DC_ONLY if dc_only else HOSTILE_DC)
        for ch in (() if dc_only else HOSTILE_CHANNELS):
            out[pb][ch] += np.float32(HOSTILE_MASSIVE)
            out[cls][0, 0, ch] += np.float32(HOSTILE_MASSIVE)
    return out


HOSTILE_DC = 2.0
HOSTILE_DC_ONLY = 4.0
HOSTILE_MASSIVE = 100.0
HOSTILE_CHANNELS = (77, 500)


# --------------------------------------------------------------------------------------
# Real checkpoints (a local HF model_dir; never fetched - there is no network)
# --------------------------------------------------------------------------------------

_V4_TO_V5 = [  # transformers-4.x ViT spelling -> 5.x (TF/conversion_mapping.py:338-346)
    (r"^encoder\.encoder\.layer\.(\d+)\.attention\.attention\.query\.", r"encoder.layers.\1.attention.q_proj."),
    (r"^encoder\.encoder\.layer\.(\d+)\.attention\.attention\.key\.", r"encoder.layers.\1.attention.k_proj."),
    (r"^encoder\.encoder\.layer\.(\d+)\.attention\.attention\.value\.", r"encoder.layers.\1.attention.v_proj."),
    (r"^encoder\.encoder\.layer\.(\d+)\.attention\.output\.dense\.", r"encoder.layers.\1.attention.o_proj."),
    (r"^encoder\.encoder\.layer\.(\d+)\.intermediate\.dense\.", r"encoder.layers.\1.mlp.fc1."),
    (r"^encoder\.encoder\.layer\.(\d+)\.output\.dense\.", r"encoder.layers.\1.mlp.fc2."),
    (r"^encoder\.encoder\.layer\.(\d+)\.layernorm_", r"encoder.layers.\1.layernorm_"),
]


def canonical_name(key: str) -> str:
    for pat, rep in _V4_TO_V5:
        new = re.sub(pat, rep, key)
        if new != key:
            return new
    return key


def spec_from_hf_config(cfg: dict) -> ModelSpec:
    """Build a ModelSpec from a VisionEncoderDecoder ``config.json`` dict, asserting the
    structural assumptions the kernels make instead of assuming them."""
    enc, dec = cfg["encoder"], cfg["decoder"]
    if enc.get("hidden_size", 768) != dec.get("hidden_size", 768):
        raise ValueError("encoder/decoder widths differ: enc_to_dec_proj is not supported")
    for side, c in (("encoder", enc), ("decoder", dec)):
        if c.get("hidden_act", "gelu") != "gelu":
            raise ValueError(f"{side} hidden_act must be exact-erf 'gelu'")
    if enc.get("num_attention_heads", 12) != dec.get("num_attention_heads", 12):
        raise ValueError("encoder/decoder head counts differ")
    spec = ModelSpec(
        image_size=enc.get("image_size", 224), patch_size=enc.get("patch_size", 16),
        hidden=enc.get("hidden_size", 768), enc_layers=enc.get("num_hidden_layers", 12),
        dec_layers=dec.get("num_hidden_layers", 2), heads=enc.get("num_attention_heads", 12),
        ffn=enc.get("intermediate_size", 3072), vocab=dec["vocab_size"],
        max_pos=dec.get("max_position_embeddings", 512), type_vocab=dec.get("type_vocab_size", 2),
        ln_eps=float(enc.get("layer_norm_eps", 1e-12)),
        max_len=int(cfg.get("max_length", dec.get("max_length", 300)) or 300),
        start_id=int(cfg.get("decoder_start_token_id", 2)),
        eos_id=int(cfg.get("eos_token_id", dec.get("eos_token_id", 3)) or 3),
        pad_id=int(cfg.get("pad_token_id", dec.get("pad_token_id", 0)) or 0),
    )
    if float(dec.get("layer_norm_eps", 1e-12)) != spec.ln_eps:
        raise ValueError("encoder/decoder layer_norm_eps differ")
    if dec.get("intermediate_size", 3072) != spec.ffn:
        raise ValueError("encoder/decoder FFN widths differ")
    # The published checkpoint's config.json is recalled to carry generation defaults of its training script
    # (num_beams=4, no_repeat_ngram_size=3, length_penalty=2.0, early_stopping=true) [RECALL]; BASELINE.json's
    # north_star fixes GREEDY decode, so the engine decodes greedily and says so once (warn, not fail).
    non_greedy = {}
    for k, dflt in (("num_beams", 1), ("no_repeat_ngram_size", 0), ("length_penalty", 1.0), ("do_sample", False),
                    ("repetition_penalty", 1.0), ("num_beam_groups", 1)):
        v = cfg.get(k, dec.get(k, dflt))
        if v is not None and v != dflt:
            non_greedy[k] = v
    if non_greedy:
        import warnings
        warnings.warn("checkpoint config asks for non-greedy generation " + repr(non_greedy) +
                      "; this engine implements greedy decode (generate(do_sample=False, num_beams=1)) and ignores it",
                      RuntimeWarning, stacklevel=2)
        spec = replace(spec, ignored_generation=tuple(sorted(non_greedy.items())))
    return spec


def _read_state_dict(model_dir: str) -> Dict[str, np.ndarray]:
    """Every floating-point tensor of the checkpoint as float32 (integer buffers such as the 4.x
    ``decoder.bert.embeddings.position_ids`` are not parameters and are dropped)."""
    st = os.path.join(model_dir, "model.safetensors")
    if os.path.exists(st):
        try:
            from safetensors.numpy import load_file
            raw = load_file(st)
            return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in raw.items() if np.issubdtype(v.dtype, np.floating)}
        except (TypeError, ValueError, KeyError):       # bf16 storage: numpy has no such dtype, go through torch
            from safetensors.torch import load_file as load_torch
            return {k: v.float().numpy() for k, v in load_torch(st).items() if v.is_floating_point()}
    pt = os.path.join(model_dir, "pytorch_model.bin")
    if not os.path.exists(pt):
        raise FileNotFoundError(f"neither model.safetensors nor pytorch_model.bin in {model_dir}")
    import torch
    sd = torch.load(pt, map_location="cpu", weights_only=True)
    return {k: v.float().numpy() for k, v in sd.items() if hasattr(v, "is_floating_point") and v.is_floating_point()}


def load_checkpoint(model_dir: str) -> Tuple[ModelSpec, Dict[str, np.ndarray]]:
    """Read ``config.json`` (+ ``generation_config.json``) and ``model.safetensors`` (or ``pytorch_model.bin``)
    from a local directory and return (spec, canonical float32 weights).  Accepts the transformers-4.x key
    spelling the published checkpoint was saved with and the 5.x one (``TF/conversion_mapping.py:338-346``), a
    tied or untied vocabulary projection (``TF/models/bert/modeling_bert.py:825-828``) and either spelling of
    its bias (``decoder.cls.predictions.bias`` / ``...decoder.bias``)."""
    with open(os.path.join(model_dir, "config.json"), "r", encoding="utf-8") as f:
        cfg = json.load(f)
    gen = os.path.join(model_dir, "generation_config.json")
    if os.path.exists(gen):        # generation defaults moved there in later transformers releases; config.json wins
        with open(gen, "r", encoding="utf-8") as f:
            for k, v in json.load(f).items():
                cfg.setdefault(k, v)
    spec = spec_from_hf_config(cfg)
    w = {canonical_name(k): v for k, v in _read_state_dict(model_dir).items()}
    lm_w = "decoder.cls.predictions.decoder.weight"
    lm_b = "decoder.cls.predictions.decoder.bias"
    if lm_w not in w:  # tied head
        w[lm_w] = w["decoder.bert.embeddings.word_embeddings.weight"].copy()
    if lm_b not in w:
        w[lm_b] = w["decoder.cls.predictions.bias"]
    wanted = {name for name, _, _ in tensor_table(spec)}
    w = {k: v for k, v in w.items() if k in wanted}      # pooler, duplicated bias, buffers: not consumed
    check_weights(w, spec)
    return spec, w


def check_weights(w: Dict[str, np.ndarray], spec: ModelSpec = DEFAULT_SPEC) -> None:
    for name, shape, _ in tensor_table(spec):
        if name not in w:
            raise KeyError(f"missing tensor {name}")
        if tuple(w[name].shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(w[name].shape)} != {tuple(shape)}")
        if w[name].dtype != np.float32:
            raise TypeError(f"{name}: dtype {w[name].dtype} != float32")
