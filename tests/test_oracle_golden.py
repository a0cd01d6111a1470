"""CPU: the oracle (oracle/) against the golden vectors generated from the HuggingFace
transformers modules (tests/golden/make_goldens.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights
from oracle import pil_ops
from oracle.mocr_oracle import Oracle, pixel_lut

TOL = 2e-5  # fp32 rounding-order differences between eager attention and HF's SDPA kernel


def crops(seed, n):
    return np.random.RandomState(seed).randint(0, 256, size=(n, 224, 224), dtype=np.uint8)


@pytest.fixture(scope="module")
def oracle0():
    return Oracle(synthetic_weights(0), DEFAULT_SPEC)


@pytest.fixture(scope="module")
def enc0(oracle0):
    pv = oracle0.preprocess_gray(crops(1234, 4))
    return oracle0.encode(pv, return_layers=True)


def test_pixel_lut_matches_hf_preprocessing(golden_dir, oracle0):
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    pv = oracle0.preprocess_gray(crops(1234, 1))
    np.testing.assert_array_equal(pv[0, :, ::37, :].numpy(), g["pixel_values_img0_rows"])
    assert pixel_lut()[0] == -1.0 and pixel_lut()[255] == 1.0


def test_pil_restatement_against_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    rs = np.random.RandomState(99)
    lut = pixel_lut()
    for i in range(5):
        h, w = g[f"size_{i}"]
        rgb = rs.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
        gray = pil_ops.preprocess_rgb_to_gray224(rgb)
        np.testing.assert_array_equal(gray, g[f"gray224_{i}"])
        np.testing.assert_array_equal(lut[gray][::56, :], g[f"pv_rows_{i}"])


def test_pil_restatement_against_pillow():
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(7)
    for (h, w) in [(224, 224), (10, 300), (448, 448), (223, 225), (1000, 30), (17, 19), (225, 224)]:
        rgb = rs.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
        img = Image.fromarray(rgb, mode="RGB")
        np.testing.assert_array_equal(pil_ops.rgb_to_l(rgb), np.asarray(img.convert("L")))
        want = np.asarray(img.convert("L").resize((224, 224), Image.BILINEAR))
        np.testing.assert_array_equal(pil_ops.preprocess_rgb_to_gray224(rgb), want)


def test_encoder_against_golden(golden_dir, enc0):
    g = np.load(os.path.join(golden_dir, "encoder_seed0.npz"))
    out, layers = enc0
    rows = g["tok_rows"]
    for name, got in (("embed_rows", layers[0]), ("layer1_rows", layers[1]), ("layer12_rows", layers[12]),
                      ("final_rows", out)):
        d = np.abs(got[:, rows, :].numpy() - g[name]).max()
        assert d <= TOL * max(1.0, np.abs(g[name]).max()), (name, d)
    np.testing.assert_allclose(out.double().sum(dim=(1, 2)).numpy(), g["final_sum"], rtol=0, atol=2e-2)
    np.testing.assert_allclose(out.double().abs().sum(dim=(1, 2)).numpy(), g["final_abs_sum"], rtol=1e-6)


def test_greedy_ids_and_logits_against_golden(golden_dir, oracle0, enc0):
    g = np.load(os.path.join(golden_dir, "decoder_seed0.npz"))
    ids, logits = oracle0.generate(enc0[0], max_len=48, return_logits=True)
    np.testing.assert_array_equal(ids.astype(np.int32), g["ids_len48"])
    assert np.abs(logits[:, :, g["vocab_cols"]] - g["logits_cols"]).max() <= TOL * 10
    assert np.abs(logits.max(-1) - g["logits_max"]).max() <= TOL * 10
    # the decisions the golden pins are not near-ties
    assert g["logits_top2_gap"].min() > 20 * TOL


def test_greedy_full_length_against_golden(golden_dir, oracle0, enc0):
    g = np.load(os.path.join(golden_dir, "decoder_seed0.npz"))
    ids = oracle0.generate(enc0[0][:2], max_len=300)
    assert ids.shape == (2, 300)
    np.testing.assert_array_equal(ids.astype(np.int32), g["ids_len300"])


def test_early_eos_padding_rule_against_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "early_eos_seed1.npz"))
    o = Oracle(synthetic_weights(1, eos_bias=1.1), DEFAULT_SPEC)
    ids = o.recognize_ids(crops(4321, 6), max_len=300)
    np.testing.assert_array_equal(ids.astype(np.int32), g["ids"])
    # rows that hit EOS are padded with pad_id afterwards, and the loop stopped when the last row finished
    assert ids.shape[1] < 300 and (ids[:, -1] != 0).any()


def test_teacher_forced_logits_equal_free_running(oracle0, enc0):
    ids, logits = oracle0.generate(enc0[0][:2], max_len=12, return_logits=True)
    _, tf_logits = oracle0.generate(enc0[0][:2], return_logits=True, forced_ids=ids[:, :-1])
    np.testing.assert_allclose(tf_logits, logits, rtol=0, atol=1e-6)


@pytest.mark.parametrize("kind,fname", [(True, "hostile_seed3.npz"), ("dc", "hostile_dc_seed3.npz")])
def test_oracle_on_hostile_residual_streams(golden_dir, kind, fname):
    """r04: the oracle against transformers on weights that give the encoder the residual-stream statistics of trained ViTs
    (synthetic_weights(3, hostile=...): LayerNorm gains U(0.2, 3), DC offsets, massive-activation channels) - the goldens
    the bf16 shortcuts are judged by (tests/test_gpu_hostile_stats.py)."""
    g = np.load(os.path.join(golden_dir, fname))
    o = Oracle(synthetic_weights(3, hostile=kind), DEFAULT_SPEC)
    enc = o.encode(o.preprocess_gray(crops(2024, 8)))
    ref = g["final_rows"]
    assert np.abs(enc.numpy()[:, g["tok_rows"], :] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    np.testing.assert_allclose(enc.double().sum(dim=(1, 2)).numpy(), g["final_sum"], rtol=0, atol=2e-3 * np.abs(ref).max())
    ids, logits = o.generate(enc, max_len=24, return_logits=True)
    np.testing.assert_array_equal(np.asarray(ids)[:, :24], g["ids_len24"])
    assert np.abs(logits[:, :23][:, :, g["vocab_cols"]] - g["logits_cols"][:, :23]).max() <= 2e-4
    # what makes the streams hostile, as transformers saw it: recorded, not re-derived
    assert g["stream_row_mean_over_info_std"][0] > 5 and (fname != "hostile_seed3.npz" or g["stream_channel_dc_max"].max() > 100)
