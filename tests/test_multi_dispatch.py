"""CPU: the main-process dispatcher (manga_ocr/multi.py) with gloo and a deterministic stand-in engine: the parent
owns the queue and DEALS chunks to whichever child has room, the children decode and keep their rows, ONE all-gather
among the children, child 0 hands every row back - order, ragged crop sizes, BGR flag, page/region remapping, two
caller threads at once, a slow child, a crop that always fails, a child that dies and a child that keeps failing."""
import os
import threading
import time

import numpy as np
import pytest

from manga_ocr.multi import MultiGpuEngine, ShardError, deal_sizes


class _Spec:
    max_len = 300


class FakeEngine:
    """ids are a pure function of the pixels (and of the flags), so a dropped, duplicated or mis-ordered row shows.
    Behaviours (factory_args): slow_rank / slow_s: that child sleeps per crop (a row's cost depends on who decodes it);
    die_rank: that child's process exits when it meets a crop of height 17; broken_rank: every call of that child raises."""
    spec = _Spec()

    def __init__(self, rank, args):
        self.rank, self.args = rank, args

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
        a = self.args
        if a.get("broken_rank") == self.rank:
            raise RuntimeError("HIP error 700: the engine refuses further calls")
        if any(im.shape[0] == 13 for im in images):
            raise ValueError("crop of height 13 is cursed")
        if a.get("die_rank") == self.rank and any(im.shape[0] == 17 for im in images):
            os._exit(3)
        if a.get("slow_rank") == self.rank:
            time.sleep(a.get("slow_s", 0.01) * len(images))
        out = [self.row(im, 1000 if bgr else 0) for im in images]
        return np.stack([o[0] for o in out]), np.array([o[1] for o in out], np.int32)

    def recognize_regions(self, pages, regions, bgr=True):
        out = [self.row(pages[p][y:y + h, x:x + w], 7) for p, x, y, w, h in regions]
        return np.stack([o[0] for o in out]), np.array([o[1] for o in out], np.int32)


def fake_factory(rank, device, args):
    return FakeEngine(rank, args)


def _crops(seed, n, lo=20, hi=60):
    rs = np.random.RandomState(seed)
    return [rs.randint(0, 256, size=(rs.randint(lo, hi), rs.randint(lo, hi), 3) if i % 3 else (rs.randint(lo, hi), 31), dtype=np.uint8)
            for i in range(n)]


def _want(crops, bgr=False):
    w = [FakeEngine.row(c, 1000 if bgr else 0) for c in crops]
    return np.stack([x[0] for x in w]), np.array([x[1] for x in w], np.int32)


@pytest.mark.parametrize("world", [2, 3])
def test_parent_deals_a_ragged_queue_and_gets_every_row_back_in_order(world):
    crops = _crops(world, 11)
    eng = MultiGpuEngine(list(range(world)), factory=fake_factory, backend="gloo", min_chunk=2, max_chunk=4)
    try:
        for bgr in (False, True):
            ids, lens = eng.recognize_images(crops, bgr)
            wi, wl = _want(crops, bgr)
            np.testing.assert_array_equal(ids, wi)
            np.testing.assert_array_equal(lens, wl)
            assert eng.stats["mode"] == "allgather" and sum(eng.stats["chunks"].values()) >= 3
        one_ids, one_lens = eng.recognize_images(crops[:1])          # fewer crops than workers: children with nothing to hand over
        np.testing.assert_array_equal(one_ids[0], FakeEngine.row(crops[0], 0)[0])
        assert eng.recognize_images([])[0].shape == (0, 300)
        # pages + regions: a child gets only the pages its rectangles touch, with remapped indices
        rs = np.random.RandomState(5)
        pages = [rs.randint(0, 256, size=(80, 90, 3), dtype=np.uint8) for _ in range(4)]
        regs = [(3, 5, 5, 20, 10), (0, 1, 2, 30, 40), (3, 0, 0, 9, 9), (2, 10, 10, 50, 50), (1, 4, 4, 8, 60)]
        rids, rlens = eng.recognize_regions(pages, regs)
        want = [FakeEngine.row(pages[p][y:y + h, x:x + w], 7) for p, x, y, w, h in regs]
        np.testing.assert_array_equal(rids, np.stack([w[0] for w in want]))
        np.testing.assert_array_equal(rlens, [w[1] for w in want])
    finally:
        eng.close()


def test_a_crop_that_always_fails_costs_its_own_row_only():
    """src/core/workers.py:241-244: an exception costs one job, the loop goes on.  The failing chunk is re-dealt to the
    other child, fails again, and is halved until the cursed crop stands alone: ShardError names row 5 and carries
    every other row; the dispatcher keeps serving (same children, collective intact)."""
    crops = _crops(1, 12)
    crops[5] = np.zeros((13, 20), np.uint8)
    eng = MultiGpuEngine([0, 1], factory=fake_factory, backend="gloo", min_chunk=4, max_chunk=8)
    try:
        with pytest.raises(ShardError, match="cursed") as ei:
            eng.recognize_images(crops)
        e = ei.value
        assert sorted(e.failed) == [5] and e.lens[5] == -1
        good = [i for i in range(12) if i != 5]
        wi, wl = _want([crops[i] for i in good])
        np.testing.assert_array_equal(e.ids[good], wi)
        np.testing.assert_array_equal(e.lens[good], wl)
        assert eng.alive == [0, 1] and eng.stats["redealt"] >= 2
        ids2, lens2 = eng.recognize_images(crops[:4])
        np.testing.assert_array_equal(ids2, _want(crops[:4])[0])
        assert eng.stats["mode"] == "allgather"
    finally:
        eng.close()


def test_two_caller_threads_share_one_handle():
    """src/ui/main_window.py:608-611, 9801: the recogniser is called from many threads without a lock.  Two threads
    submit different queues at the same moment, many times over; every call gets its own rows."""
    eng = MultiGpuEngine([0, 1], factory=fake_factory, backend="gloo", min_chunk=1, max_chunk=3)
    errs = []

    def caller(seed):
        try:
            for k in range(6):
                crops = _crops(100 * seed + k, 5 + (seed + k) % 4)
                ids, lens = eng.recognize_images(crops, bgr=bool(seed & 1))
                wi, wl = _want(crops, bool(seed & 1))
                assert (ids == wi).all() and (lens == wl).all()
        except BaseException as exc:      # noqa: BLE001
            errs.append(exc)

    try:
        ts = [threading.Thread(target=caller, args=(s,)) for s in (1, 2, 3)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
        assert not errs, errs
        assert not any(t.is_alive() for t in ts)
    finally:
        eng.close()


def test_a_slow_child_takes_fewer_chunks():
    """Pull-based dealing (src/ui/main_window.py:4329-4335: workers POP jobs): child 0 needs 15 ms per crop, child 1
    none - with static halves the job would take 40 x 15 ms / 2; dealt on demand, child 1 ends up with most chunks."""
    crops = _crops(7, 40, 20, 30)
    eng = MultiGpuEngine([0, 1], factory=fake_factory, factory_args=dict(slow_rank=0, slow_s=0.015), backend="gloo", min_chunk=2, max_chunk=2)
    try:
        t0 = time.perf_counter()
        ids, lens = eng.recognize_images(crops)
        dt = time.perf_counter() - t0
        np.testing.assert_array_equal(ids, _want(crops)[0])
        ch = eng.stats["chunks"]
        assert ch[1] > ch[0] and ch[0] >= 1, ch
        assert dt < 40 * 0.015 / 2, f"{dt:.3f} s: no faster than static halves"
    finally:
        eng.close()


def test_a_child_that_dies_loses_nothing():
    """a rank's shard is re-queued on the survivors.  Child 1 exits (os._exit) on the crop of height 17:
    its outstanding and its already decoded chunks go back to the queue, child 0 decodes them, the rows are written
    straight into the result block (the collective is not entered with a rank missing), and the handle keeps serving
    on the survivor."""
    crops = _crops(3, 16)
    crops[2] = np.full((17, 25, 3), 7, np.uint8)       # chunk [2, 4) is the first one dealt to child 1 (breadth-first dealing)
    eng = MultiGpuEngine([0, 1], factory=fake_factory, factory_args=dict(die_rank=1), backend="gloo", min_chunk=2, max_chunk=2)
    try:
        ids, lens = eng.recognize_images(crops)
        wi, wl = _want(crops)
        np.testing.assert_array_equal(ids, wi)
        np.testing.assert_array_equal(lens, wl)
        assert eng.alive == [0] and eng.stats["mode"] == "direct" and eng.stats["lost"][0][0] == 1
        ids, _ = eng.recognize_images(crops[:8])
        np.testing.assert_array_equal(ids, _want(crops[:8])[0])
        assert eng.stats["mode"] == "direct"
    finally:
        eng.close()


def test_a_child_whose_engine_keeps_failing_is_dropped():
    """A poisoned engine (every call raises, as after a HIP fault): each of its chunks is decoded by the other child,
    and after four failures in a row it is no longer dealt anything."""
    crops = _crops(4, 24)
    eng = MultiGpuEngine([0, 1], factory=fake_factory, factory_args=dict(broken_rank=1), backend="gloo", min_chunk=2, max_chunk=2)
    try:
        ids, lens = eng.recognize_images(crops)
        np.testing.assert_array_equal(ids, _want(crops)[0])
        assert eng.alive == [0] and "failed" in eng.stats["lost"][0][1]
    finally:
        eng.close()


def test_deal_sizes_cover_the_queue():
    cases = ((10_000, 8, 4096, 64), (5, 3, 4, 2), (1, 2, 8, 4), (129, 2, 64, 64), (0, 4, 8, 2), (100_000, 8, 5120, 64), (100, 8, 4096, 64))
    for policy in ("equal", "guided"):
        for n, world, mx, mn in cases:
            ch = deal_sizes(n, world, mx, mn, policy=policy)
            assert (not ch and n == 0) or (ch[0][0] == 0 and ch[-1][1] == n)
            assert all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
            sizes = [b - a for a, b in ch]
            assert all(0 < s <= mx for s in sizes)
            assert sizes == sorted(sizes, reverse=True) or policy == "guided" and sizes[:-1] == sorted(sizes[:-1], reverse=True)
            if policy == "guided":
                assert all(s >= min(mn, n) for s in sizes[:-1])
            else:
                assert max(sizes, default=0) - min(sizes, default=0) <= 1           # whole rounds of equal chunks
    # the 10,000-crop queue of BASELINE configs[3] on 8 GPUs whose engines take 2 x 2048 rows: the static shards, dealt
    assert [b - a for a, b in deal_sizes(10_000, 8, 4096, 64, policy="equal")] == [1250] * 8
    # ... three rounds when the queue outgrows what the children hold at once; never thinner than min_chunk while that covers it
    assert len(deal_sizes(100_000, 8, 5120, 64, policy="equal")) == 24
    assert [b - a for a, b in deal_sizes(100, 8, 4096, 64, policy="equal")] == [50, 50]
    sizes = [b - a for a, b in deal_sizes(10_000, 8, 4096, 64, policy="guided")]
    assert sizes[0] == 625 and sizes[-2] >= 64 and len(sizes) >= 16
    with pytest.raises(ValueError):
        deal_sizes(10, 2, 4, 2, policy="static")


def failing_factory(rank, device, args):
    if rank == 1:
        raise RuntimeError("no GPU for you")
    return FakeEngine(rank, args)


def test_a_child_that_cannot_build_its_engine_fails_the_constructor():
    with pytest.raises(RuntimeError, match="no GPU for you"):
        MultiGpuEngine([0, 1], factory=failing_factory, backend="gloo")
