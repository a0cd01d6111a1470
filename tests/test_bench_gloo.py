"""bench.py's own N > 1 path - ranks, shards, the one all-gather, max over ranks, ONE JSON line on rank 0 - rehearsed on
CPU at world size 2 (gloo) with the stand-in engine bench.py carries for exactly this (MOCR_BENCH_FAKE_ENGINE=1): the
driver launches the real thing on 8 GPUs with the same command line, and nothing else in this container can run it."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, extra):
    env = dict(os.environ, MOCR_BENCH_FAKE_ENGINE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout              # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def _key(rank, b):
    g = np.random.RandomState(1234 + rank).randint(0, 256, size=(b, 224, 224), dtype=np.uint8)
    return g.reshape(b, -1).astype(np.int64).sum(1) % 6000


def test_weak_scaling_line_at_world_2():
    B, K, L = 8, 3, 16
    d = _run(2, ["--steps", str(K), "--warmup", "1", "--batch", str(B), "--max-len", str(L)])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == K and d["warmup"] == 1
    assert d["value"] > 0 and abs(d["value"] - 2 * B * K / (d["ms_per_step"] * 1e-3 * K)) < 1e-6 * d["value"]
    assert d["config"]["global_batch"] == 2 * B and d["config"]["rccl_world_size"] == 2 and d["config"]["parallelism"] == "dp2"
    # every rank's K steps arrived, rank 0's block first: (2 + key + 3) per row
    want = sum(int((5 + _key(r, B)).sum()) * K for r in range(2))
    assert d["gathered_shape"] == [2 * K, B, L] and d["gathered_checksum"] == want


def _strong_want(chunks, B):
    k = _key(0, B)                                        # strong mode: one set of B crops, every chunk replays it from its first row
    return sum(5 + int(k[i % B]) + 3 for lo, hi in chunks for i in range(hi - lo))      # ids 2, key, 3 and the length column (3)


def test_strong_scaling_queue_is_dealt_in_equal_chunks_at_world_2():
    """--queue goes through the product dispatcher's chunking (manga_ocr.multi.deal_sizes): 37 crops on two ranks whose engines
    take 2 x 8 rows are three chunks of 13 + 12 + 12 (no thinner than the 16 rows a child holds), pulled from the shared counter;
    ragged last batches."""
    sys.path.insert(0, os.path.join(ROOT, "manga-ocr_amd"))
    from manga_ocr.multi import deal_sizes
    B, L, Q = 8, 16, 37
    d = _run(2, ["--queue", str(Q), "--batch", str(B), "--max-batch", "8", "--lanes", "2", "--max-len", str(L), "--steps", "2", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["queue"] == Q
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] - Q) < 1e-6 * Q      # value = the WHOLE queue over the slowest rank's time
    chunks = deal_sizes(Q, 2, 16)
    assert [b - a for a, b in chunks] == [13, 12, 12]
    deal = d["dealing"]
    assert deal["policy"] == "equal" and deal["chunks"] == 3 and deal["chunk_rows"] == [13, 12, 12]
    assert sum(deal["chunks_pulled_per_rank"]) == 3 and len(deal["chunks_pulled_per_rank"]) == 2      # every chunk pulled exactly once
    assert d["gathered_shape"] == [Q, L + 1] and d["gathered_checksum"] == _strong_want(chunks, B)


def test_strong_scaling_queue_guided_chunks_at_world_2():
    sys.path.insert(0, os.path.join(ROOT, "manga-ocr_amd"))
    from manga_ocr.multi import deal_sizes
    B, L, Q = 8, 16, 150
    d = _run(2, ["--queue", str(Q), "--deal", "guided", "--batch", str(B), "--max-batch", "16", "--lanes", "2", "--max-len", str(L), "--steps", "2", "--warmup", "1"])
    chunks = deal_sizes(Q, 2, 32, policy="guided")
    assert d["dealing"]["policy"] == "guided" and d["dealing"]["chunks"] == len(chunks) and sum(d["dealing"]["chunks_pulled_per_rank"]) == len(chunks)
    assert d["gathered_shape"] == [Q, L + 1] and d["gathered_checksum"] == _strong_want(chunks, B)
