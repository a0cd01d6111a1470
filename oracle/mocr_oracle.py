"""CPU ORACLE for the Manga-OCR recogniser hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (manga-ocr_amd/) never does and fails loudly without its HIP
library.

What it restates
----------------
The reference (irazawa/Manga-OCR) calls ``self.manga_ocr_reader(pil_img)`` at
``src/ui/main_window.py:9801``.  The arithmetic behind that call is NOT in the
reference tree: it lives in the un-vendored, un-pinned pip dependency ``manga-ocr``
which wraps HuggingFace ``transformers`` (``VisionEncoderDecoderModel`` = ViT-B/16
encoder + 2-layer BERT decoder, greedy ``generate(max_length=300)``).  This file is a
plain-torch fp32 restatement of that published algorithm, following transformers
5.15.0 (the version in the build container; paths below are relative to the
``transformers`` package, "TF/"):

  preprocessing      TF/image_processing_backends.py:619-647, TF/image_transforms.py:89-124, 385-440
  ViT embeddings     TF/models/vit/modeling_vit.py:42-161
  ViT layer          TF/models/vit/modeling_vit.py:164-286 ; final LayerNorm :385
  BERT embeddings    TF/models/bert/modeling_bert.py:53-108
  BERT self-attn     TF/models/bert/modeling_bert.py:139-203 ; cross-attn :206-279
  BERT layer         TF/models/bert/modeling_bert.py:354-416 ; LM head :466-496
  greedy loop        TF/generation/utils.py:2783-2973 ; stopping TF/generation/stopping_criteria.py:58-77, 534-581

Pinning
-------
The reference holds no tests, golden vectors or fixtures for this path (SURVEY.md
§4, §8c), so by the reference's own data parity is UNPINNED.  The oracle is pinned
instead against outputs of the transformers modules themselves, run in the build
container on this repo's deterministic synthetic weights: tests/golden/make_goldens.py
generated tests/golden/*.npz, and tests/test_oracle_golden.py checks this file against
them (encoder states and logits to <= 2e-5, token ids exactly).

Attention is written in the "eager" form softmax(q k^T * scale) v (TF .../modeling_vit.py
:164-189); HF's default SDPA kernel differs from it only in fp32 rounding order.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _t(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a))


def pixel_lut() -> np.ndarray:
    """uint8 -> normalised fp32, exactly as HF computes it: float32(float64(u)*(1/255)),
    then (x - float32(0.5)) / float32(0.5)   (TF/image_transforms.py:118-122, 437)."""
    x = (np.arange(256).astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
    return ((x - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


class Oracle:
    """fp32 CPU restatement.  ``weights`` uses the canonical names of
    manga_ocr.weights.tensor_table (transformers-5.x state_dict spelling)."""

    def __init__(self, weights: Dict[str, np.ndarray], spec=None, threads: Optional[int] = None):
        if spec is None:
            from types import SimpleNamespace
            spec = SimpleNamespace(image_size=224, patch_size=16, hidden=768, enc_layers=12, dec_layers=2,
                                   heads=12, ffn=3072, vocab=6144, max_pos=512, ln_eps=1e-12, max_len=300,
                                   start_id=2, eos_id=3, pad_id=0)
        self.spec = spec
        self.w = {k: _t(v).float() for k, v in weights.items()}
        self.H = spec.heads
        self.D = spec.hidden
        self.dh = spec.hidden // spec.heads
        self.eps = spec.ln_eps
        if threads:
            torch.set_num_threads(threads)

    # ---------------------------------------------------------------- preprocessing
    def preprocess_gray(self, gray_u8: np.ndarray) -> torch.Tensor:
        """[N,224,224] uint8 luminance (what the model sees after convert('L').convert('RGB')
        and the identity resize) -> pixel_values fp32 [N,3,224,224]."""
        lut = _t(pixel_lut())
        x = lut[_t(gray_u8.astype(np.int64))]
        return x[:, None, :, :].expand(-1, 3, -1, -1).contiguous()

    # ---------------------------------------------------------------- encoder
    def _ln(self, x, prefix):
        return F.layer_norm(x, (self.D,), self.w[prefix + ".weight"], self.w[prefix + ".bias"], self.eps)

    def _attn(self, q, k, v):
        # q [B,H,Sq,dh], k/v [B,H,Sk,dh] ; eager_attention_forward
        s = torch.matmul(q, k.transpose(-1, -2)) * (self.dh ** -0.5)
        p = torch.softmax(s, dim=-1, dtype=torch.float32)
        return torch.matmul(p, v)

    def _heads(self, x):
        B, S, _ = x.shape
        return x.view(B, S, self.H, self.dh).transpose(1, 2)

    def _merge(self, x):
        B, H, S, dh = x.shape
        return x.transpose(1, 2).reshape(B, S, H * dh)

    def embed_patches(self, pixel_values: torch.Tensor) -> torch.Tensor:
        w = self.w
        e = "encoder.embeddings."
        x = F.conv2d(pixel_values, w[e + "patch_embeddings.projection.weight"],
                     w[e + "patch_embeddings.projection.bias"], stride=self.spec.patch_size)
        x = x.flatten(2).transpose(1, 2)                                  # [B,196,768]
        cls = w[e + "cls_token"].expand(x.shape[0], -1, -1)
        x = torch.cat((cls, x), dim=1)
        return x + w[e + "position_embeddings"]

    def encoder_layer(self, x: torch.Tensor, i: int) -> torch.Tensor:
        w = self.w
        p = f"encoder.layers.{i}."
        h = self._ln(x, p + "layernorm_before")
        q = self._heads(F.linear(h, w[p + "attention.q_proj.weight"], w[p + "attention.q_proj.bias"]))
        k = self._heads(F.linear(h, w[p + "attention.k_proj.weight"], w[p + "attention.k_proj.bias"]))
        v = self._heads(F.linear(h, w[p + "attention.v_proj.weight"], w[p + "attention.v_proj.bias"]))
        a = self._merge(self._attn(q, k, v))
        a = F.linear(a, w[p + "attention.o_proj.weight"], w[p + "attention.o_proj.bias"])
        x = x + a
        h = self._ln(x, p + "layernorm_after")
        h = F.gelu(F.linear(h, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"]))
        h = F.linear(h, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
        return x + h

    @torch.no_grad()
    def encode(self, pixel_values: torch.Tensor, return_layers: bool = False):
        x = self.embed_patches(pixel_values)
        layers = [x]
        for i in range(self.spec.enc_layers):
            x = self.encoder_layer(x, i)
            layers.append(x)
        out = self._ln(x, "encoder.layernorm")       # pooler is computed by HF but unused
        return (out, layers) if return_layers else out

    # ---------------------------------------------------------------- decoder
    @torch.no_grad()
    def cross_kv(self, enc: torch.Tensor) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        out = []
        for i in range(self.spec.dec_layers):
            p = f"decoder.bert.encoder.layer.{i}.crossattention.self."
            k = self._heads(F.linear(enc, self.w[p + "key.weight"], self.w[p + "key.bias"]))
            v = self._heads(F.linear(enc, self.w[p + "value.weight"], self.w[p + "value.bias"]))
            out.append((k, v))
        return out

    def _dec_embed(self, ids: torch.Tensor, pos: int) -> torch.Tensor:
        d = "decoder.bert.embeddings."
        e = self.w[d + "word_embeddings.weight"][ids] + self.w[d + "token_type_embeddings.weight"][0]
        e = e + self.w[d + "position_embeddings.weight"][pos]
        return self._ln(e, d + "LayerNorm")

    def _dec_layer_step(self, x, i, self_kv, ckv):
        """x [B,1,768]; self_kv: list holding (K,V) [B,H,t,dh] or None; post-LN residual blocks."""
        w = self.w
        p = f"decoder.bert.encoder.layer.{i}."
        a = p + "attention."
        q = self._heads(F.linear(x, w[a + "self.query.weight"], w[a + "self.query.bias"]))
        k = self._heads(F.linear(x, w[a + "self.key.weight"], w[a + "self.key.bias"]))
        v = self._heads(F.linear(x, w[a + "self.value.weight"], w[a + "self.value.bias"]))
        if self_kv[i] is not None:
            k = torch.cat((self_kv[i][0], k), dim=2)
            v = torch.cat((self_kv[i][1], v), dim=2)
        self_kv[i] = (k, v)
        ctx = self._merge(self._attn(q, k, v))          # query length 1: the causal mask is vacuous
        h = F.linear(ctx, w[a + "output.dense.weight"], w[a + "output.dense.bias"])
        x = self._ln(h + x, a + "output.LayerNorm")
        c = p + "crossattention."
        q = self._heads(F.linear(x, w[c + "self.query.weight"], w[c + "self.query.bias"]))
        ctx = self._merge(self._attn(q, ckv[i][0], ckv[i][1]))
        h = F.linear(ctx, w[c + "output.dense.weight"], w[c + "output.dense.bias"])
        x = self._ln(h + x, c + "output.LayerNorm")
        h = F.gelu(F.linear(x, w[p + "intermediate.dense.weight"], w[p + "intermediate.dense.bias"]))
        h = F.linear(h, w[p + "output.dense.weight"], w[p + "output.dense.bias"])
        return self._ln(h + x, p + "output.LayerNorm")

    def _lm_head(self, x):
        c = "decoder.cls.predictions."
        h = F.gelu(F.linear(x, self.w[c + "transform.dense.weight"], self.w[c + "transform.dense.bias"]))
        h = self._ln(h, c + "transform.LayerNorm")
        return F.linear(h, self.w[c + "decoder.weight"], self.w[c + "decoder.bias"])

    @torch.no_grad()
    def decode_step(self, tok: torch.Tensor, pos: int, self_kv, ckv) -> torch.Tensor:
        """One KV-cached decoder forward: tok [B] int64 at position ``pos`` -> fp32 logits [B,V]."""
        x = self._dec_embed(tok[:, None], pos)
        for i in range(self.spec.dec_layers):
            x = self._dec_layer_step(x, i, self_kv, ckv)
        return self._lm_head(x)[:, 0, :].float()

    @torch.no_grad()
    def generate(self, enc: torch.Tensor, max_len: Optional[int] = None, return_logits: bool = False,
                 forced_ids: Optional[np.ndarray] = None):
        """Greedy loop (GenerationMixin._sample with do_sample=False, num_beams=1).

        Returns ids int64 [B, L] (L <= max_len; rows padded with pad_id after EOS, exactly
        like ``generate``), and optionally the per-step fp32 logits [B, L-1, V].
        ``forced_ids`` [B, T] teacher-forces the inputs (logits of every step are returned,
        the ids returned are the forced ones)."""
        sp = self.spec
        max_len = max_len or sp.max_len
        B = enc.shape[0]
        ckv = self.cross_kv(enc)
        self_kv = [None] * sp.dec_layers
        ids = torch.full((B, 1), sp.start_id, dtype=torch.int64)
        unfinished = torch.ones(B, dtype=torch.int64)
        logits_all = []
        t = 0
        while True:
            tok = ids[:, -1] if forced_ids is None else _t(forced_ids[:, t].astype(np.int64))
            logits = self.decode_step(tok, t, self_kv, ckv)
            if return_logits:
                logits_all.append(logits)
            if forced_ids is not None:
                t += 1
                if t >= forced_ids.shape[1]:
                    ids = _t(forced_ids.astype(np.int64))
                    break
                continue
            nxt = torch.argmax(logits, dim=-1)
            nxt = nxt * unfinished + sp.pad_id * (1 - unfinished)          # utils.py:2929
            ids = torch.cat((ids, nxt[:, None]), dim=1)                    # :2932
            done = (nxt == sp.eos_id) | (ids.shape[1] >= max_len)          # stopping_criteria
            unfinished = unfinished & (~done).long()
            t += 1
            if int(unfinished.max()) == 0:
                break
        if return_logits:
            return ids.numpy(), torch.stack(logits_all, dim=1).numpy()
        return ids.numpy()

    # ---------------------------------------------------------------- whole path
    @torch.no_grad()
    def recognize_ids(self, gray_u8: np.ndarray, max_len: Optional[int] = None) -> np.ndarray:
        enc = self.encode(self.preprocess_gray(gray_u8))
        return self.generate(enc, max_len=max_len)


def pad_ids(ids: np.ndarray, max_len: int, pad_id: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """HF-style [B,L] ids -> the C-ABI's fixed [B,max_len] block + per-row lengths
    (length = index of the first EOS + 1, or L when the row never finished)."""
    B, L = ids.shape
    out = np.full((B, max_len), pad_id, dtype=np.int32)
    out[:, :L] = ids
    lens = np.full(B, L, dtype=np.int32)
    return out, lens


def row_lengths(ids: np.ndarray, eos_id: int = 3) -> np.ndarray:
    """Per-row token count up to and including the first EOS (whole row if none)."""
    B, L = ids.shape
    lens = np.full(B, L, dtype=np.int32)
    for b in range(B):
        hit = np.nonzero(ids[b, 1:] == eos_id)[0]
        if hit.size:
            lens[b] = hit[0] + 2
    return lens
