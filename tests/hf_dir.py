"""Test helper: write a synthetic HuggingFace ``model_dir`` shaped like the published ``kha-white/manga-ocr-base``
checkpoint [RECALL]: transformers-4.x key spelling for the ViT part (``TF/conversion_mapping.py:338-346`` is the
4.x -> 5.x rename table), tied vocabulary projection (no ``decoder.cls.predictions.decoder.weight`` in the file,
``TF/models/bert/modeling_bert.py:825-828``), the bias under ``decoder.cls.predictions.bias``, the unused ViT pooler,
the int64 ``position_ids`` buffer older checkpoints carry, a ``config.json`` with the training script's generation
defaults (num_beams=4 ...), ``preprocessor_config.json`` and a ``vocab.txt``.  No download, no reference file."""
import json
import os
import re

import numpy as np

from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights

_V5_TO_V4 = [
    (r"^encoder\.layers\.(\d+)\.attention\.q_proj\.", r"encoder.encoder.layer.\1.attention.attention.query."),
    (r"^encoder\.layers\.(\d+)\.attention\.k_proj\.", r"encoder.encoder.layer.\1.attention.attention.key."),
    (r"^encoder\.layers\.(\d+)\.attention\.v_proj\.", r"encoder.encoder.layer.\1.attention.attention.value."),
    (r"^encoder\.layers\.(\d+)\.attention\.o_proj\.", r"encoder.encoder.layer.\1.attention.output.dense."),
    (r"^encoder\.layers\.(\d+)\.mlp\.fc1\.", r"encoder.encoder.layer.\1.intermediate.dense."),
    (r"^encoder\.layers\.(\d+)\.mlp\.fc2\.", r"encoder.encoder.layer.\1.output.dense."),
    (r"^encoder\.layers\.(\d+)\.layernorm_", r"encoder.encoder.layer.\1.layernorm_"),
]


def v4_name(key: str) -> str:
    for pat, rep in _V5_TO_V4:
        new = re.sub(pat, rep, key)
        if new != key:
            return new
    return key


def vocab_tokens(size=6144):
    """Five special tokens, some half-width ASCII / digits / katakana (so post_process has work to do), a
    '##' word piece, then CJK ideographs."""
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    toks += list("abcXYZ019!?.") + ["ｱ", "ｶ", "ﾞ", "ﾊ", "ﾟ", "…", "・", "##ー"]
    toks += [chr(0x4E00 + i) for i in range(size - len(toks))]
    return toks


def write_hf_dir(path, seed=0, spec=DEFAULT_SPEC, num_beams=4, **synth):
    os.makedirs(path, exist_ok=True)
    w = synthetic_weights(seed, spec, **synth)
    sd = {}
    for k, v in w.items():
        if k == "decoder.cls.predictions.decoder.weight":
            continue                                        # tied: not stored
        if k == "decoder.cls.predictions.decoder.bias":
            sd["decoder.cls.predictions.bias"] = v          # where 4.x BertLMPredictionHead kept it
            continue
        sd[v4_name(k)] = v
    rs = np.random.RandomState(99)
    sd["encoder.pooler.dense.weight"] = (0.02 * rs.standard_normal((spec.hidden, spec.hidden))).astype(np.float32)
    sd["encoder.pooler.dense.bias"] = np.zeros(spec.hidden, np.float32)
    sd["decoder.bert.embeddings.position_ids"] = np.arange(spec.max_pos, dtype=np.int64)[None]
    from safetensors.numpy import save_file
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    cfg = {
        "architectures": ["VisionEncoderDecoderModel"], "model_type": "vision-encoder-decoder",
        "decoder_start_token_id": spec.start_id, "eos_token_id": spec.eos_id, "pad_token_id": spec.pad_id,
        "max_length": spec.max_len, "num_beams": num_beams, "no_repeat_ngram_size": 3, "length_penalty": 2.0,
        "early_stopping": True, "tie_word_embeddings": False,
        "encoder": {"model_type": "vit", "hidden_size": spec.hidden, "num_hidden_layers": spec.enc_layers,
                    "num_attention_heads": spec.heads, "intermediate_size": spec.ffn, "hidden_act": "gelu",
                    "image_size": spec.image_size, "patch_size": spec.patch_size, "num_channels": 3,
                    "layer_norm_eps": spec.ln_eps, "qkv_bias": True},
        "decoder": {"model_type": "bert", "vocab_size": spec.vocab, "hidden_size": spec.hidden,
                    "num_hidden_layers": spec.dec_layers, "num_attention_heads": spec.heads,
                    "intermediate_size": spec.ffn, "hidden_act": "gelu", "max_position_embeddings": spec.max_pos,
                    "type_vocab_size": spec.type_vocab, "layer_norm_eps": spec.ln_eps, "is_decoder": True,
                    "add_cross_attention": True, "tie_word_embeddings": True},
    }
    with open(os.path.join(path, "config.json"), "w", encoding="utf-8") as f:
        json.dump(cfg, f)
    with open(os.path.join(path, "preprocessor_config.json"), "w", encoding="utf-8") as f:
        json.dump({"do_resize": True, "size": spec.image_size, "resample": 2, "do_normalize": True,
                   "image_mean": [0.5, 0.5, 0.5], "image_std": [0.5, 0.5, 0.5]}, f)
    with open(os.path.join(path, "vocab.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(vocab_tokens(spec.vocab)) + "\n")
    return w
