"""Data-parallel crop-job queue over the GPUs of one node (BASELINE configs[3]).

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests).  Crops are independent, so the queue is partitioned into contiguous shards with no
data-path collective; the only exchange is ONE all-gather of the fixed-width token-id block
(int32 [n_local_max, max_len]) and the per-row lengths, after which every rank (and in particular
the "main process", rank 0, which owns the application) holds all decoded rows in queue order.
The reference has no counterpart: its queue is a Python list popped by QThreads
(src/ui/main_window.py:4329-4335) - this is what replaces it when the job list is large.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition: the first n % world ranks get one extra crop."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_rows(ids: Optional[np.ndarray], lens: Optional[np.ndarray], n: int, max_len: int = 300, group=None,
                device: Optional[str] = None) -> Tuple[np.ndarray, np.ndarray]:
    """THE exchange step: this rank's decoded shard (rows [lo, hi) of a queue of n, `shard_bounds`) -> the whole queue
    on every rank, with ONE all-gather of a fixed-size int32 [n_local_max, max_len + 1] block (the last column
    carries the row length)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n, world, rank)
    n_max = max(1, -(-n // world))
    block = np.zeros((n_max, max_len + 1), dtype=np.int32)
    if hi > lo:
        block[:hi - lo, :max_len] = ids
        block[:hi - lo, max_len] = lens
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.from_numpy(block).to(device)
    out = torch.empty((world * n_max, max_len + 1), dtype=mine.dtype, device=device)
    dist.all_gather_into_tensor(out, mine, group=group)           # the one exchange step
    out = out.cpu().numpy().reshape(world, n_max, max_len + 1)
    ids_all = np.zeros((n, max_len), dtype=np.int32)
    lens_all = np.zeros(n, dtype=np.int32)
    for r in range(world):
        a, b = shard_bounds(n, world, r)
        ids_all[a:b] = out[r, :b - a, :max_len]
        lens_all[a:b] = out[r, :b - a, max_len]
    return ids_all, lens_all


def recognize_sharded(gray: np.ndarray, recognize: Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray]],
                      max_len: int = 300, group=None, device: Optional[str] = None) -> Tuple[np.ndarray, np.ndarray]:
    """gray: the WHOLE job queue, uint8 [N,224,224], identical on every rank (or only this rank's
    shard is touched - rows outside [lo,hi) are never read).  ``recognize`` maps a shard to
    (ids int32 [n,max_len], lens int32 [n]) - normally ``Engine.recognize``.
    Returns (ids [N,max_len], lens [N]) on every rank."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return recognize(gray)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = int(gray.shape[0])
    lo, hi = shard_bounds(n, world, rank)
    ids = lens = None
    if hi > lo:
        ids, lens = recognize(gray[lo:hi])
    return gather_rows(ids, lens, n, max_len, group, device)


def texts_from_ids(vocab, ids: np.ndarray, lens: Sequence[int]) -> List[str]:
    from .text import ids_to_text
    return [ids_to_text(vocab, ids[i, :lens[i]]) for i in range(len(lens))]
