"""CPU: host logic - C-ABI library loads and exports every declared symbol, weight spec, text
post-processing, and that the product path fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest

import __graft_entry__ as entry
from manga_ocr import _capi, text
from manga_ocr.weights import (DEFAULT_SPEC, canonical_name, check_weights, spec_from_hf_config, synthetic_weights,
                               tensor_table)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    entry.build()
    return _capi.load_library()


def test_library_exports_every_symbol_of_the_header(lib):
    hdr = open(os.path.join(ROOT, "include", "mocr.h")).read()
    declared = set(re.findall(r"\b(mocr_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"mocr_engine", "mocr_config", "mocr_kernel_stat"}
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.mocr_abi_version() == 2


def test_config_struct_matches_header():
    hdr = open(os.path.join(ROOT, "include", "mocr.h")).read()
    body = hdr[hdr.index("typedef struct mocr_config {"):hdr.index("} mocr_config;")]
    fields = re.findall(r"^\s*(?:int32_t|float)\s+(\w+);", body, flags=re.M)
    assert fields == [f[0] for f in _capi.MocrConfig._fields_]


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from manga_ocr.engine import Engine
    with pytest.raises(_capi.MocrError):
        Engine(synthetic_weights(0), DEFAULT_SPEC, dtype="bf16", max_batch=1)


def test_bad_config_is_rejected_without_touching_a_gpu(lib):
    import ctypes as C
    cfg = _capi.MocrConfig(struct_size=4)
    h = C.c_void_p()
    assert lib.mocr_create(C.byref(cfg), C.byref(h)) == -1 and not h.value
    assert lib.mocr_set_tensor(None, b"x", None, None, 0) == -1


def test_synthetic_weights_are_deterministic_and_complete():
    a, b = synthetic_weights(0), synthetic_weights(0)
    check_weights(a)
    assert len(a) == len(tensor_table()) == 4 + 12 * 16 + 2 + 5 + 2 * 26 + 6
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    assert sum(v.size for v in a.values()) == 111_005_952 - 590_592 + 6144 * 768  # HF count, minus the unused pooler, plus the untied head
    c = synthetic_weights(1)
    assert not np.array_equal(a["encoder.layernorm.weight"], c["encoder.layernorm.weight"])
    # frozen stream: first values of the first tensor never change
    np.testing.assert_allclose(a["encoder.embeddings.cls_token"].ravel()[:3],
                               0.02 * np.random.RandomState(0).standard_normal(3), rtol=1e-6)
    e = synthetic_weights(1, eos_bias=1.1)
    d = e["decoder.cls.predictions.decoder.bias"] - c["decoder.cls.predictions.decoder.bias"]
    assert d[3] == pytest.approx(1.1) and np.count_nonzero(d) == 1


def test_checkpoint_key_renames_and_config_checks():
    assert canonical_name("encoder.encoder.layer.3.attention.attention.query.weight") == "encoder.layers.3.attention.q_proj.weight"
    assert canonical_name("encoder.encoder.layer.11.output.dense.bias") == "encoder.layers.11.mlp.fc2.bias"
    assert canonical_name("encoder.encoder.layer.0.layernorm_before.weight") == "encoder.layers.0.layernorm_before.weight"
    assert canonical_name("decoder.bert.encoder.layer.1.output.dense.bias") == "decoder.bert.encoder.layer.1.output.dense.bias"
    cfg = {"encoder": {}, "decoder": {"vocab_size": 6144, "num_hidden_layers": 2}, "decoder_start_token_id": 2,
           "eos_token_id": 3, "pad_token_id": 0, "max_length": 300}
    assert spec_from_hf_config(cfg) == DEFAULT_SPEC
    # the published config carries its training script's beam-search defaults [RECALL]; north_star fixes greedy:
    # the loader says so once and goes on
    with pytest.warns(RuntimeWarning, match="greedy"):
        beam = spec_from_hf_config({**cfg, "num_beams": 4, "no_repeat_ngram_size": 3, "length_penalty": 2.0})
        assert beam == DEFAULT_SPEC          # the model is the same; what the engine will NOT do is kept for the caller to read
        assert dict(beam.ignored_generation) == {"num_beams": 4, "no_repeat_ngram_size": 3, "length_penalty": 2.0}
        assert DEFAULT_SPEC.ignored_generation == ()
    with pytest.raises(ValueError):
        spec_from_hf_config({**cfg, "encoder": {"hidden_act": "gelu_new"}})


def test_load_checkpoint_end_to_end_from_a_synthetic_hf_directory(tmp_path):
    """4.x key spelling, tied LM head, bias under its 4.x name, pooler + position_ids present, num_beams=4 in
    config.json: every line of load_checkpoint runs, and the result equals the generator's canonical weights."""
    from hf_dir import write_hf_dir
    from manga_ocr.weights import load_checkpoint
    d = str(tmp_path / "manga-ocr-base")
    want = write_hf_dir(d, seed=3, eos_bias=0.5)
    from safetensors.numpy import load_file
    stored = load_file(os.path.join(d, "model.safetensors"))
    assert "decoder.cls.predictions.decoder.weight" not in stored and "encoder.encoder.layer.0.attention.attention.query.weight" in stored
    with pytest.warns(RuntimeWarning, match="greedy"):
        spec, got = load_checkpoint(d)
    assert spec == DEFAULT_SPEC
    assert set(got) == set(want)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    v = text.Vocab.from_file(text.find_vocab(d))
    assert len(v) == spec.vocab


def test_post_process_known_answers():
    # spec-derived cases ([RECALL] of manga_ocr.post_process + jaconv.h2z tables)
    assert text.post_process("ｱ ｲ ｳ") == "アイウ"
    assert text.post_process("ｶﾞｷﾞ ﾊﾟ") == "ガギパ"
    assert text.post_process("abc 123!?") == "ａｂｃ１２３！？"
    assert text.post_process("あ…い") == "あ．．．い"
    assert text.post_process("あ・・・い・う") == "あ．．．い・う"
    assert text.post_process("え . . 。") == "え．．。"
    assert text.post_process('"a\'b" \\ `') == "”ａ’ｂ”￥‘"          # jaconv's typographic partners [RECALL]
    assert text._h2z_tables("ｳﾞｧ ﾎﾟ ｶ", ascii=True, digit=True) == "ヴァ　ポ　カ"
    assert len(text._HALF_ASCII) == len(text._FULL_ASCII) and len(text._HALF_KANA) == len(text._FULL_KANA)


def test_vocab_decode_skips_special_tokens_and_joins_wordpieces():
    v = text.Vocab(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "日", "本", "##語", "a"])
    assert v.decode([2, 5, 6, 7, 3, 0, 0]) == "日 本語"
    assert text.ids_to_text(v, [2, 5, 6, 7, 8, 3, 0]) == "日本語ａ"
    s = text.Vocab.synthetic(6144)
    assert len(s) == 6144 and text.ids_to_text(s, [2, 5, 6, 3]) == "一丁"


def test_fast_ids_to_text_equals_decode_then_post_process():
    """ids_to_text's precomputed per-token path must be the plain composition tokenizer.decode -> post_process for every
    row: word pieces ('##'), whitespace inside tokens, '…', dot runs across token seams, half-width kana with and without
    (semi-)voiced marks, ids outside the table."""
    from hf_dir import vocab_tokens
    edge = text.Vocab(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", ".", "･", "・", "…", "ｶ", "ﾞ", "##.", "a", "1", "##ｶ", "ﾊ", "ﾟ",
                       "あ", "　", "##", "x y"] + [chr(0x4E00 + i) for i in range(60)])
    for v in (text.Vocab.synthetic(6144), text.Vocab(vocab_tokens()), edge):
        rs = np.random.RandomState(len(v))
        ids = rs.randint(-2, len(v) + 2, size=(300, 120))
        ids[:, :60] = rs.randint(0, min(40, len(v)), size=(300, 60))
        for row in ids:
            assert text.ids_to_text(v, row) == text.post_process(v.decode(row))
        assert text.ids_to_text(v, []) == "" and text.ids_to_text(v, [2, 3, 0]) == ""
