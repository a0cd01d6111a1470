"""-m gpu: the device luminance conversion + Pillow-exact BILINEAR resize (csrc/preprocess.h) through the C
ABI (mocr_preprocess / mocr_recognize_images) - bit-exact against the oracle's Pillow restatement, against
the golden planes Pillow itself wrote (tests/golden/preprocess.npz), and token-identical to the path that
resizes on the host."""
import os

import numpy as np
import pytest

from gpu_util import engine, report
from oracle import pil_ops

pytestmark = pytest.mark.gpu

SIZES = [(224, 224), (10, 300), (448, 448), (223, 225), (1000, 30), (17, 19), (225, 224), (224, 100), (64, 224),
         (1, 1), (3, 2), (511, 513), (32, 512), (2000, 1500)]


def test_device_preprocess_matches_pillow_golden(golden_dir):
    eng = engine("bf16")
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    rs = np.random.RandomState(99)          # the generator's stream (tests/golden/make_goldens.py, case C)
    imgs = []
    for i in range(5):
        h, w = g[f"size_{i}"]
        imgs.append(rs.randint(0, 256, size=(h, w, 3), dtype=np.uint8))
    got = eng.preprocess(imgs)
    for i in range(5):
        np.testing.assert_array_equal(got[i], g[f"gray224_{i}"])
    report("device preprocess == Pillow golden planes (5 sizes), bit-exact")


@pytest.mark.parametrize("channels", [3, 1])
def test_device_preprocess_matches_oracle_all_sizes(channels):
    eng = engine("bf16")
    rs = np.random.RandomState(7 + channels)
    imgs = [rs.randint(0, 256, size=(h, w, 3) if channels == 3 else (h, w), dtype=np.uint8) for (h, w) in SIZES]
    got = eng.preprocess(imgs)            # one ragged batch: every size in one launch pair
    for im, plane, (h, w) in zip(imgs, got, SIZES):
        want = pil_ops.preprocess_rgb_to_gray224(im) if channels == 3 else pil_ops.resize_bilinear_u8(im, 224, 224)
        np.testing.assert_array_equal(plane, want, err_msg=f"size {h}x{w}")
    report(f"device preprocess == oracle for {len(SIZES)} sizes, channels={channels}, bit-exact")


def test_device_preprocess_extremes_and_strides():
    """Constant images stay constant (the coefficients of every output pixel sum to 2^22 up to rounding that the
    8-bit result absorbs for 0 and 255), and rows with padding between them are read with their stride."""
    eng = engine("bf16")
    for v in (0, 255):
        out = eng.preprocess([np.full((37, 401), v, np.uint8), np.full((600, 5, 3), v, np.uint8)])
        assert (out == v).all()
    rs = np.random.RandomState(3)
    wide = rs.randint(0, 256, size=(90, 200, 3), dtype=np.uint8)
    view = wide[:, :150]                    # non-contiguous rows: Engine.preprocess packs them; the C ABI takes row_stride
    np.testing.assert_array_equal(eng.preprocess([view])[0], pil_ops.preprocess_rgb_to_gray224(np.ascontiguousarray(view)))
    with pytest.raises(Exception):
        eng.preprocess([np.zeros((4, 4, 2), np.uint8)])


@pytest.mark.parametrize("bgr", [False, True])
def test_device_rotation_is_bit_exact_with_rot90_then_pillow(bgr):
    """the reference's orientation-only rotation (src/core/workers.py:320-326,
    src/ui/main_window.py:9787-9795: cv2.ROTATE_90_CLOCKWISE for "Vertical" + landscape, _COUNTERCLOCKWISE for
    "Horizontal" + portrait) folded into the resize kernel's source addressing: mocr_image.rotate on the crop as it lies
    in memory == np.rot90 on the host followed by the oracle's Pillow restatement, plane for plane, for every case of the
    rule, RGB / BGR / L, odd sizes, a 224-sized side (skipped pass) and a crop that is a strided view into a page."""
    from manga_ocr.queue_worker import ROTATE_90_CCW, ROTATE_90_CW, ROTATE_NONE, orient_crop, rotation_code
    eng = engine("bf16")
    rs = np.random.RandomState(21 + bgr)
    page = rs.randint(0, 256, size=(700, 900, 3), dtype=np.uint8)
    crops = [rs.randint(0, 256, size=(h, w, 3), dtype=np.uint8) for (h, w) in ((90, 300), (300, 90), (224, 51), (51, 224), (1, 7), (333, 224))]
    crops += [page[100:260, 40:500], page[5:600, 700:820]]                 # views with the page's row stride
    crops += [rs.randint(0, 256, size=(64, 200), dtype=np.uint8)]          # L
    for orient in ("Vertical", "Horizontal", "Auto-Detect"):
        rot = [rotation_code(c.shape[0], c.shape[1], orient) for c in crops]
        got = eng.preprocess(crops, bgr=bgr, rotate=rot)
        for c, r, plane in zip(crops, rot, got):
            host = orient_crop(c, orient)                                  # the reference's own order: rotate first ...
            assert (r == ROTATE_NONE) == (host is c)
            if host.ndim == 3:
                rgb = np.ascontiguousarray(host[..., ::-1]) if bgr else np.ascontiguousarray(host)
                want = pil_ops.preprocess_rgb_to_gray224(rgb)              # ... then BGR -> RGB, convert('L'), resize
            else:
                want = pil_ops.resize_bilinear_u8(np.ascontiguousarray(host), 224, 224)
            np.testing.assert_array_equal(plane, want, err_msg=f"{c.shape} {orient} rot={r}")
        assert ROTATE_90_CW in rot or ROTATE_90_CCW in rot or orient == "Auto-Detect"
    with pytest.raises(Exception):
        eng.preprocess(crops[:1], rotate=[3])
    report(f"device rotation (mocr_image.rotate) == np.rot90 + Pillow restatement, {len(crops)} crops x 3 orientation settings, bgr={bgr}")


def test_recognize_images_equals_host_resize_path():
    """Crops of mixed sizes through mocr_recognize_images == the same crops resized by the oracle's Pillow
    restatement on the host and sent through mocr_recognize: the planes are bit-identical, so are the ids."""
    eng = engine("bf16")
    rs = np.random.RandomState(11)
    sizes = [(224, 224), (120, 333), (500, 90), (64, 64), (300, 300)]
    imgs = [rs.randint(0, 256, size=(h, w, 3), dtype=np.uint8) for (h, w) in sizes]
    ids_dev, len_dev = eng.recognize_images(imgs)
    grays = np.stack([pil_ops.preprocess_rgb_to_gray224(im) for im in imgs])
    ids_host, len_host = eng.recognize(grays)
    np.testing.assert_array_equal(ids_dev, ids_host)
    np.testing.assert_array_equal(len_dev, len_host)
    report("recognize_images (device resize) ids == host-resize path, 5 mixed sizes")
