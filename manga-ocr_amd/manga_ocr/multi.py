"""Main-process dispatcher over the GPUs of one node: ``MangaOcr(devices=[0..7])``.

The reference is ONE process that owns the crop-job queue (``src/ui/main_window.py:4286-4335``) and the page loop
(``src/core/workers.py:448-482``).  To spread that queue over the 8 MI355X of a node without turning the application
into an SPMD program, the parent process - which never touches a GPU - starts one FRESH child process per device
(``multiprocessing`` spawn context: a new interpreter, so no child inherits GPU state and nothing that has
initialised HIP is ever re-exec'ed), each child builds its own engine, and per job:

    parent: packs the crops (or pages) once into a shared-memory block, sends every child the descriptor list
    child r: decodes its contiguous shard [lo_r, hi_r) of the queue (``shard.shard_bounds``), no data-path collective
    children: ONE all-gather of the fixed-width id block among themselves (``shard.gather_rows``; torch.distributed,
              backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests)
    child 0: writes the gathered [N, max_len + 1] int32 rows into the parent's result block

so the parent gets all decoded rows in queue order from one place, as north_star words it ("RCCL all-gather ... back
to the main process").  Weights are replicated (222 MB bf16 per GPU).
"""
from __future__ import annotations

import os
import socket
import traceback
from multiprocessing import shared_memory
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .shard import gather_rows, shard_bounds

_TIMEOUT = float(os.environ.get("MANGA_OCR_WORKER_TIMEOUT", "600"))


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def default_engine_factory(rank: int, device: int, args: dict):
    """Runs in the child: the HIP engine on this child's device."""
    from .engine import Engine
    from .weights import DEFAULT_SPEC, load_checkpoint, synthetic_weights
    if args.get("synthetic_seed") is not None:
        spec, weights = DEFAULT_SPEC, synthetic_weights(args["synthetic_seed"], **args.get("synthetic_kwargs", {}))
    else:
        spec, weights = load_checkpoint(args["model_dir"])
    return Engine(weights, spec, dtype=args.get("dtype", "bf16"), device=device, max_batch=args.get("max_batch", 1024),
                  lanes=args.get("lanes", 2), flags=args.get("flags", 0))


def _views(buf: np.ndarray, descs) -> List[np.ndarray]:
    out = []
    for off, h, w, ch in descs:
        a = buf[off:off + h * w * ch]
        out.append(a.reshape(h, w) if ch == 1 else a.reshape(h, w, ch))
    return out


def _worker_main(rank: int, world: int, device: int, port: int, backend: str, factory: Callable, factory_args: dict, conn) -> None:
    """Child process: build the engine, join the group, serve jobs until told to stop."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        tdev = "cpu"
        if backend == "nccl":
            torch.cuda.set_device(device)
            tdev = f"cuda:{device}"
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        engine = factory(rank, device, factory_args)
        max_len = int(engine.spec.max_len)
        conn.send(("ready", rank, max_len))
    except BaseException as exc:       # noqa: BLE001 - reported to the parent, which raises
        conn.send(("error", rank, f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"))
        return
    while True:
        msg = conn.recv()
        if msg[0] == "stop":
            break
        try:
            kind, in_name, out_name, n = msg[0], msg[1], msg[2], msg[3]
            lo, hi = shard_bounds(n, world, rank)
            shm = shared_memory.SharedMemory(name=in_name)
            ids = lens = None
            err = None
            try:
                buf = np.ndarray((shm.size,), dtype=np.uint8, buffer=shm.buf)
                if hi > lo:
                    try:
                        if kind == "images":
                            descs, bgr = msg[4], msg[5]
                            ids, lens = engine.recognize_images(_views(buf, descs[lo:hi]), bgr)
                        else:                       # regions: this child's rectangles, and only the pages they touch
                            pdescs, regs, bgr = msg[4], msg[5], msg[6]
                            mine = regs[lo:hi]
                            used = sorted({r[0] for r in mine})
                            slot = {p: i for i, p in enumerate(used)}
                            pages = _views(buf, [pdescs[p] for p in used])
                            ids, lens = engine.recognize_regions(pages, [(slot[r[0]],) + tuple(r[1:]) for r in mine], bgr)
                    except BaseException as exc:   # noqa: BLE001 - the collective below must still be entered by every rank
                        err = f"{type(exc).__name__}: {exc}"
                        ids = np.zeros((hi - lo, max_len), np.int32)
                        lens = np.full(hi - lo, -1, np.int32)        # length -1 marks a failed shard
                del buf
            finally:
                shm.close()
            ids_all, lens_all = gather_rows(ids, lens, n, max_len, None, tdev)
            if rank == 0:
                out = shared_memory.SharedMemory(name=out_name)
                try:
                    o = np.ndarray((n, max_len + 1), dtype=np.int32, buffer=out.buf)
                    o[:, :max_len] = ids_all
                    o[:, max_len] = lens_all
                    del o
                finally:
                    out.close()
            conn.send(("done", rank, err))
        except BaseException as exc:       # noqa: BLE001
            conn.send(("error", rank, f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"))
            break
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
        close = getattr(engine, "close", None)
        if close:
            close()
    except BaseException:                  # noqa: BLE001
        pass


class MultiGpuEngine:
    """The parent-side handle: same ``recognize_images`` / ``recognize_regions`` surface as ``Engine``."""

    def __init__(self, devices: Sequence[int], factory: Callable = default_engine_factory, factory_args: Optional[dict] = None,
                 backend: str = "nccl"):
        import multiprocessing as mp
        if len(devices) < 1:
            raise ValueError("devices must name at least one GPU")
        self.devices = [int(d) for d in devices]
        self.world = len(self.devices)
        ctx = mp.get_context("spawn")          # fresh interpreters: no GPU state is inherited, nothing is re-exec'ed
        port = _free_port()
        self._conns, self._procs = [], []
        for r, d in enumerate(self.devices):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker_main, args=(r, self.world, d, port, backend, factory, dict(factory_args or {}), child),
                            name=f"mocr-gpu{d}", daemon=True)
            p.start()
            child.close()
            self._conns.append(parent)
            self._procs.append(p)
        self.max_len = None
        try:
            for r, c in enumerate(self._conns):
                msg = self._recv(c, r)
                self.max_len = int(msg[2])
        except BaseException:
            self.close()
            raise
        self._closed = False

    # ------------------------------------------------------------------ plumbing
    def _recv(self, conn, rank):
        if not conn.poll(_TIMEOUT):
            raise RuntimeError(f"GPU worker {rank} (device {self.devices[rank]}) did not answer within {_TIMEOUT:.0f} s")
        try:
            msg = conn.recv()
        except EOFError:
            raise RuntimeError(f"GPU worker {rank} (device {self.devices[rank]}) died") from None
        if msg[0] == "error":
            raise RuntimeError(f"GPU worker {msg[1]} failed: {msg[2]}")
        return msg

    def _run(self, n: int, payload_bytes: int, fill: Callable[[np.ndarray], None], message: Callable[[str, str], tuple]):
        L = self.max_len
        shm_in = shared_memory.SharedMemory(create=True, size=max(payload_bytes, 1))
        shm_out = shared_memory.SharedMemory(create=True, size=max(n * (L + 1) * 4, 4))
        try:
            buf = np.ndarray((shm_in.size,), dtype=np.uint8, buffer=shm_in.buf)
            fill(buf)
            del buf
            msg = message(shm_in.name, shm_out.name)
            for c in self._conns:
                c.send(msg)
            errs = []
            for r, c in enumerate(self._conns):
                done = self._recv(c, r)
                if done[2]:
                    errs.append(f"worker {r}: {done[2]}")
            if errs:
                raise RuntimeError("; ".join(errs))
            o = np.ndarray((n, L + 1), dtype=np.int32, buffer=shm_out.buf)
            ids, lens = o[:, :L].copy(), o[:, L].copy()
            del o
            return ids, lens
        finally:
            for s in (shm_in, shm_out):
                s.close()
                s.unlink()

    @staticmethod
    def _pack(arrays):
        descs, off = [], 0
        norm = []
        for im in arrays:
            a = np.ascontiguousarray(im, dtype=np.uint8)
            if a.ndim == 2:
                ch = 1
            elif a.ndim == 3 and a.shape[2] == 3:
                ch = 3
            else:
                raise ValueError("each image must be uint8 [h,w] or [h,w,3]")
            descs.append((off, a.shape[0], a.shape[1], ch))
            off += a.size
            norm.append(a)

        def fill(buf):
            for (o, h, w, ch), a in zip(descs, norm):
                buf[o:o + a.size] = a.reshape(-1)
        return descs, off, fill

    # ------------------------------------------------------------------ the hot path
    def recognize_images(self, images, bgr: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        n = len(images)
        if n == 0:
            return np.zeros((0, self.max_len), np.int32), np.zeros(0, np.int32)
        descs, size, fill = self._pack(images)
        return self._run(n, size, fill, lambda a, b: ("images", a, b, n, descs, bool(bgr)))

    def recognize_regions(self, pages, regions, bgr: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        regs = [tuple(int(v) for v in r) for r in regions]
        n = len(regs)
        if n == 0:
            return np.zeros((0, self.max_len), np.int32), np.zeros(0, np.int32)
        pdescs, size, fill = self._pack(pages)
        return self._run(n, size, fill, lambda a, b: ("regions", a, b, n, pdescs, regs, bool(bgr)))

    def close(self) -> None:
        if getattr(self, "_closed", False):
            return
        self._closed = True
        for c in self._conns:
            try:
                c.send(("stop",))
            except (OSError, BrokenPipeError):
                pass
        for p in self._procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()          # the exact child this object started
        for c in self._conns:
            c.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
