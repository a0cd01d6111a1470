"""CPU: the N>1 path - queue sharding + the one all-gather - with gloo, world_size 2 and 3, and a
deterministic stand-in for the engine (no GPU here)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from manga_ocr.shard import recognize_sharded, shard_bounds


def fake_engine(gray):
    """ids are a pure function of the crop, so any mis-ordering or dropped row is visible."""
    n = gray.shape[0]
    ids = np.zeros((n, 300), dtype=np.int32)
    lens = np.zeros(n, dtype=np.int32)
    for i in range(n):
        h = int(gray[i].astype(np.uint64).sum() % 997)
        L = 2 + h % 50
        ids[i, 0] = 2
        ids[i, 1:L - 1] = 5 + (h + np.arange(L - 2)) % 6000
        ids[i, L - 1] = 3
        lens[i] = L
    return ids, lens


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gray = np.random.RandomState(7).randint(0, 256, size=(n, 224, 224), dtype=np.uint8)
    ids, lens = recognize_sharded(gray, fake_engine, max_len=300)
    want_ids, want_lens = fake_engine(gray)
    q.put((rank, bool((ids == want_ids).all() and (lens == want_lens).all())))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 11), (3, 4), (2, 1)])
def test_sharded_queue_all_gather(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == list(range(world)) and all(ok for _, ok in res)


def test_shard_bounds_cover_the_queue_exactly():
    for n in (0, 1, 7, 64, 10_000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
