"""CPU: the main-process dispatcher (manga_ocr/multi.py) with gloo and a deterministic stand-in engine: the parent
owns the queue, N fresh children decode contiguous shards, ONE all-gather among the children, child 0 hands every
row back - order, ragged crop sizes, BGR flag, page/region remapping and error reporting are all visible."""
import numpy as np
import pytest

from manga_ocr.multi import MultiGpuEngine


class _Spec:
    max_len = 300


class FakeEngine:
    """ids are a pure function of the pixels (and of the flags), so a dropped, duplicated or mis-ordered row shows."""
    spec = _Spec()

    def __init__(self, rank):
        self.rank = rank

    @staticmethod
    def row(pixels, salt):
        h = int(np.asarray(pixels, dtype=np.uint64).sum() % 9973) + salt
        L = 2 + h % 60
        ids = np.zeros(300, np.int32)
        ids[0] = 2
        ids[1:L - 1] = 5 + (h + np.arange(L - 2)) % 6000
        ids[L - 1] = 3
        return ids, L

    def recognize_images(self, images, bgr=False):
        if any(im.shape[0] == 13 for im in images):
            raise ValueError("crop of height 13 is cursed")
        out = [self.row(im, 1000 if bgr else 0) for im in images]
        return np.stack([o[0] for o in out]), np.array([o[1] for o in out], np.int32)

    def recognize_regions(self, pages, regions, bgr=True):
        out = [self.row(pages[p][y:y + h, x:x + w], 7) for p, x, y, w, h in regions]
        return np.stack([o[0] for o in out]), np.array([o[1] for o in out], np.int32)


def fake_factory(rank, device, args):
    return FakeEngine(rank)


@pytest.mark.parametrize("world", [2, 3])
def test_parent_shards_a_ragged_queue_and_gets_every_row_back_in_order(world):
    rs = np.random.RandomState(world)
    crops = [rs.randint(0, 256, size=(rs.randint(20, 60), rs.randint(20, 60), 3) if i % 3 else (rs.randint(20, 60), 31),
                        dtype=np.uint8) for i in range(11)]
    eng = MultiGpuEngine(list(range(world)), factory=fake_factory, backend="gloo")
    try:
        for bgr in (False, True):
            ids, lens = eng.recognize_images(crops, bgr)
            want = [FakeEngine.row(c, 1000 if bgr else 0) for c in crops]
            np.testing.assert_array_equal(ids, np.stack([w[0] for w in want]))
            np.testing.assert_array_equal(lens, [w[1] for w in want])
        one_ids, one_lens = eng.recognize_images(crops[:1])          # fewer crops than workers: empty shards
        np.testing.assert_array_equal(one_ids[0], FakeEngine.row(crops[0], 0)[0])
        assert eng.recognize_images([])[0].shape == (0, 300)
        # pages + regions: a child gets only the pages its rectangles touch, with remapped indices
        pages = [rs.randint(0, 256, size=(80, 90, 3), dtype=np.uint8) for _ in range(4)]
        regs = [(3, 5, 5, 20, 10), (0, 1, 2, 30, 40), (3, 0, 0, 9, 9), (2, 10, 10, 50, 50), (1, 4, 4, 8, 60)]
        rids, rlens = eng.recognize_regions(pages, regs)
        want = [FakeEngine.row(pages[p][y:y + h, x:x + w], 7) for p, x, y, w, h in regs]
        np.testing.assert_array_equal(rids, np.stack([w[0] for w in want]))
        np.testing.assert_array_equal(rlens, [w[1] for w in want])
        # a failing shard is reported, names the worker, and the dispatcher keeps serving afterwards
        bad = crops[:5] + [np.zeros((13, 20), np.uint8)]
        with pytest.raises(RuntimeError, match="cursed"):
            eng.recognize_images(bad)
        ids2, _ = eng.recognize_images(crops[:4])
        np.testing.assert_array_equal(ids2[3], FakeEngine.row(crops[3], 0)[0])
    finally:
        eng.close()


def failing_factory(rank, device, args):
    if rank == 1:
        raise RuntimeError("no GPU for you")
    return FakeEngine(rank)


def test_a_child_that_cannot_build_its_engine_fails_the_constructor():
    with pytest.raises(RuntimeError, match="no GPU for you"):
        MultiGpuEngine([0, 1], factory=failing_factory, backend="gloo")
