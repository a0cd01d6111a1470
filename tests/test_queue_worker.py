"""CPU: the crop-job queue (manga_ocr/queue_worker.py) - FIFO completion order, micro-batching, per-job error isolation,
and the crop rules it restates from the reference's call sites."""
import threading

import numpy as np
import pytest

from manga_ocr.queue_worker import (ROTATE_90_CCW, ROTATE_90_CW, ROTATE_NONE, CropJob, CropJobQueue, bgr_to_rgb, clean_and_join,
                                    ocr_failed, orient_crop, padded_region_crop, rotation_code)


def test_orientation_rule():
    land = np.arange(2 * 5 * 3, dtype=np.uint8).reshape(2, 5, 3)       # h=2, w=5
    port = np.ascontiguousarray(land.transpose(1, 0, 2))               # h=5, w=2
    assert orient_crop(land, "Auto-Detect") is land and orient_crop(port, "Vertical") is port
    cw = orient_crop(land, "Vertical")                                  # landscape + Vertical: 90 degrees clockwise
    assert cw.shape == (5, 2, 3)
    np.testing.assert_array_equal(cw[0, 0], land[1, 0])                 # bottom-left -> top-left; top-left -> top-right
    np.testing.assert_array_equal(cw[0, 1], land[0, 0])
    ccw = orient_crop(port, "Horizontal")                               # portrait + Horizontal: counter-clockwise
    assert ccw.shape == (2, 5, 3)
    np.testing.assert_array_equal(ccw[0, 0], port[0, 1])                # top-right -> top-left
    np.testing.assert_array_equal(bgr_to_rgb(land)[..., 0], land[..., 2])


def test_rotation_code_is_the_orientation_rule():
    """The code handed to the engine (mocr_image.rotate) follows src/core/workers.py:320-326 case for case: what
    orient_crop does to the pixels on the host, the device does for these codes."""
    for (h, w) in ((2, 5), (5, 2), (4, 4)):
        a = np.arange(h * w * 3, dtype=np.uint8).reshape(h, w, 3)
        for o in ("Auto-Detect", "Vertical", "Horizontal", "whatever"):
            code = rotation_code(h, w, o)
            want = {ROTATE_NONE: a, ROTATE_90_CW: np.rot90(a, k=-1), ROTATE_90_CCW: np.rot90(a, k=1)}[code]
            np.testing.assert_array_equal(orient_crop(a, o), want)
    assert rotation_code(2, 5, "Vertical") == ROTATE_90_CW and rotation_code(5, 2, "Horizontal") == ROTATE_90_CCW
    assert rotation_code(5, 2, "Vertical") == ROTATE_NONE and rotation_code(2, 5, "Horizontal") == ROTATE_NONE


def test_oriented_recogniser_gets_the_raw_bgr_crops_and_their_orientations():
    seen, done = [], []

    def recognize_oriented(crops, orients):
        seen.append((list(crops), list(orients)))
        return [f"{c.shape[0]}x{c.shape[1]}:{o}" for c, o in zip(crops, orients)]

    q = CropJobQueue(None, lambda j, t: done.append((j.payload, t)), max_batch=4, recognize_oriented=recognize_oriented)
    jobs = [CropJob(_crop(i, 4, 9), orientation=o, payload=i) for i, o in enumerate(("Vertical", "Horizontal", "Auto-Detect"))]
    try:
        for j in jobs:
            q.submit(j)
        assert q.join(20)
    finally:
        q.close()
    assert done == [(0, "4x9:Vertical"), (1, "4x9:Horizontal"), (2, "4x9:Auto-Detect")]
    assert all(c is j.crop_bgr for (cs, _), _ in zip(seen, [0]) for c, j in zip(cs, jobs))     # handed over untouched: no rotate, no swap
    with pytest.raises(ValueError):
        CropJobQueue(None, lambda j, t: None)


def test_padded_region_crop():
    page = np.arange(100 * 200, dtype=np.uint32).reshape(100, 200)
    c = padded_region_crop(page, 50, 40, 100, 20)                       # pad = int(0.08 * 100) = 8
    assert c.shape == (36, 116) and c[0, 0] == page[32, 42]
    c = padded_region_crop(page, 0, 0, 30, 30)                          # clipped at the page border
    assert c.shape == (32, 32)
    assert padded_region_crop(page, 199, 50, 1, 1) is None              # a 1-pixel sliver is not a crop
    assert c.base is None                                               # a copy, not a view of the page


def test_text_rules():
    assert clean_and_join("  a \n b\t c ") == "a b c"
    assert ocr_failed("") and ocr_failed(" \n") and ocr_failed("x [ERROR: y]") and ocr_failed("[TESSERACT ERROR: z]")
    assert not ocr_failed("こんにちは")


def _crop(tag, h=4, w=6):
    a = np.zeros((h, w, 3), dtype=np.uint8)
    a[0, 0, 0] = tag                                                    # B channel of the BGR crop
    return a


def test_fifo_order_batches_and_error_isolation():
    calls, done, errs = [], [], []
    gate = threading.Event()

    def recognize(crops):
        gate.wait(5)
        tags = [int(c[0, 0, 2]) for c in crops]                        # after BGR -> RGB the tag sits in channel 2
        calls.append(tags)
        if any(t == 13 for t in tags):
            raise ValueError("bad crop 13")
        return [f"t{t}" for t in tags]

    q = CropJobQueue(recognize, lambda j, t: done.append((j.payload, t)), lambda j, e: errs.append((j.payload, str(e))), max_batch=8)
    try:
        for i in range(20):
            q.submit(CropJob(_crop(i), payload=i, pre_detected_text="known" if i == 5 else None))
        gate.set()
        assert q.join(20)
    finally:
        q.close()
    assert [p for p, _ in done] == [i for i in range(20) if i != 13]   # submission order, job 13 missing
    assert dict(done)[5] == "known" and dict(done)[12] == "t12" and dict(done)[14] == "t14"
    assert errs == [(13, "bad crop 13")]
    assert all(len(c) <= 8 for c in calls) and max(len(c) for c in calls) > 1          # micro-batches, never above max_batch
    assert 5 not in [t for c in calls for t in c]                       # a pre-detected text skips the recogniser
    with pytest.raises(RuntimeError):
        q.submit(CropJob(_crop(1)))
