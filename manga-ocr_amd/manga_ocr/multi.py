"""Main-process dispatcher over the GPUs of one node: ``MangaOcr(devices=[0..7])``.

The reference is ONE process that owns the crop-job queue (``src/ui/main_window.py:4286-4335``) and the page loop
(``src/core/workers.py:448-482``); its worker threads POP jobs (``main_window.py:4329-4335``), so a slow job never
idles the others, and an exception costs one job while the loop goes on (``workers.py:241-244``).  To spread that queue
over the 8 MI355X of a node without turning the application into an SPMD program, the parent process - which never
touches a GPU - starts one FRESH child process per device (``multiprocessing`` spawn context: a new interpreter, so no
child inherits GPU state and nothing that has initialised HIP is ever re-exec'ed), each child builds its own engine, and
per job:

    parent : packs the crops (or pages) once into a shared-memory block and announces the job to every child
    parent : DEALS chunks of the queue to whichever child has room (pull-based: at most two chunks outstanding per
             child, sizes shrinking towards the end of the queue - no static shards)
    child  : decodes each chunk it is dealt and keeps the rows; no data-path collective
    children: ONE all-gather of a fixed-width int32 block (ids, length, queue position) among themselves
             (torch.distributed, backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests)
    child 0: scatters the gathered rows into the parent's result block by queue position

so the parent gets all decoded rows in queue order from one place, as north_star words it ("RCCL all-gather ... back
to the main process").  Weights are replicated (222 MB bf16 per GPU).

Failures (``workers.py:241-244``):
  * a chunk whose decode raises is re-dealt to ANOTHER child; if it fails again it is bisected, so that a bad crop
    costs exactly its own row: the call then raises :class:`ShardError`, which carries every other row's result;
  * a child that dies (or stops answering) is dropped: its outstanding chunks go back to the queue for the survivors,
    and from then on the survivors write their rows straight into the result block - the collective is never entered
    with a rank missing; a chunk that has killed two children is reported failed instead of being dealt a third time;
  * a failure inside the exchange step itself cannot be repaired: every child this object started is terminated, the
    handle is marked broken and every later call raises at once (fresh children belong in a new ``MultiGpuEngine``).

The handle is thread-safe: the reference calls one shared recogniser from up to 15 threads without a lock
(``main_window.py:608-611, 9801``); here a lock serialises whole jobs, so two callers can never interleave their
messages on the per-child pipes.
"""
from __future__ import annotations

import datetime
import os
import socket
import threading
import time
import traceback
from collections import deque
from multiprocessing import connection as mp_connection
from multiprocessing import shared_memory
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

_TIMEOUT = float(os.environ.get("MANGA_OCR_WORKER_TIMEOUT", "600"))
_MIN_CHUNK = int(os.environ.get("MANGA_OCR_MIN_CHUNK", "64"))       # rows: below this a chunk no longer amortises a decode
_DEPTH = 2                                                            # chunks a child may hold: one decoding, one queued behind it


class ShardError(RuntimeError):
    """Some rows of a job could not be decoded.  ``failed`` maps queue position -> message; ``ids`` / ``lens`` hold the
    results of every other row (failed rows: pad ids, length -1)."""

    def __init__(self, failed: Dict[int, str], ids: np.ndarray, lens: np.ndarray):
        self.failed, self.ids, self.lens = dict(failed), ids, lens
        first = sorted(failed)[:3]
        super().__init__(f"{len(failed)} of {len(lens)} crops failed; first: " + "; ".join(f"row {r}: {failed[r]}" for r in first))


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


# ------------------------------------------------------------------------------------------------ child
def _decode_chunk(engine, job, buf, lo, hi):
    if job["kind"] == "images":
        rot = job.get("rotate")
        if rot is None:                          # (engines and stand-ins that do not know the argument are not handed it)
            return engine.recognize_images(_views(buf, job["descs"][lo:hi]), job["bgr"])
        return engine.recognize_images(_views(buf, job["descs"][lo:hi]), job["bgr"], rot[lo:hi])
    # regions: this chunk's rectangles, and only the pages they touch
    mine = job["regs"][lo:hi]
    used = sorted({r[0] for r in mine})
    slot = {p: i for i, p in enumerate(used)}
    pages = _views(buf, [job["descs"][p] for p in used])
    return engine.recognize_regions(pages, [(slot[r[0]],) + tuple(r[1:]) for r in mine], job["bgr"])


def _worker_main(rank: int, world: int, device: int, port: int, backend: str, factory: Callable, factory_args: dict, conn) -> None:
    """Child process: build the engine, join the group, serve jobs until told to stop."""
    engine = None
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        tdev = "cpu"
        # an explicit timeout: a collective that is short of one rank must END (the parent then tears everything down)
        tmo = datetime.timedelta(seconds=max(30.0, min(_TIMEOUT, 1800.0)))
        if backend == "nccl":
            torch.cuda.set_device(device)
            tdev = f"cuda:{device}"
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device), timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
        engine = factory(rank, device, factory_args)
        max_len = int(engine.spec.max_len)
        conn.send(("ready", rank, max_len))
    except BaseException as exc:       # noqa: BLE001 - reported to the parent, which raises
        conn.send(("error", rank, f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"))
        return
    job, shm_in, buf, rows = None, None, None, []

    def drop_job():
        nonlocal job, shm_in, buf, rows
        buf = None
        if shm_in is not None:
            shm_in.close()
        job, shm_in, rows = None, None, []

    while True:
        try:
            msg = conn.recv()
        except EOFError:               # the parent is gone
            break
        try:
            if msg[0] == "stop":
                break
            if msg[0] == "job":        # ("job", id, header): a new queue; nothing is decoded yet
                drop_job()
                job = dict(msg[2], id=msg[1])
                shm_in = shared_memory.SharedMemory(name=job["shm_in"])
                buf = np.ndarray((shm_in.size,), dtype=np.uint8, buffer=shm_in.buf)
            elif msg[0] == "chunk":    # ("chunk", id, lo, hi): decode rows [lo, hi) of the queue and keep them
                _, jid, lo, hi = msg
                err = None
                try:
                    if job is None or jid != job["id"]:
                        raise RuntimeError("chunk of an unknown job")
                    ids, lens = _decode_chunk(engine, job, buf, lo, hi)
                    rows.append((lo, hi, np.asarray(ids, np.int32), np.asarray(lens, np.int32)))
                except Exception as exc:   # noqa: BLE001 - costs this chunk only (src/core/workers.py:241-244)
                    err = f"{type(exc).__name__}: {exc}"
                conn.send(("chunk_done", rank, jid, lo, hi, err))
            elif msg[0] == "gather":   # ("gather", id, mode, n_max): hand the kept rows over, then forget the job
                _, jid, mode, n_max = msg
                n = job["n"]
                mine = np.full((max(n_max, 1), max_len + 2), -1, np.int32)     # [ids | length | queue position]
                k = 0
                for lo, hi, ids, lens in rows:
                    m = hi - lo
                    mine[k:k + m, :max_len] = ids
                    mine[k:k + m, max_len] = lens
                    mine[k:k + m, max_len + 1] = np.arange(lo, hi)
                    k += m
                if mode == "allgather":
                    t = torch.from_numpy(mine).to(tdev)
                    out = torch.empty((world * mine.shape[0], max_len + 2), dtype=t.dtype, device=tdev)
                    dist.all_gather_into_tensor(out, t)                         # THE exchange step
                    got = out.cpu().numpy() if rank == 0 else None
                else:                  # "direct": a rank is missing - no collective; every survivor writes its own rows
                    got = mine
                if got is not None:
                    shm_out = shared_memory.SharedMemory(name=job["shm_out"])
                    try:
                        o = np.ndarray((n, max_len + 1), dtype=np.int32, buffer=shm_out.buf)
                        sel = got[:, max_len + 1] >= 0
                        o[got[sel, max_len + 1]] = got[sel, :max_len + 1]
                        del o
                    finally:
                        shm_out.close()
                drop_job()
                conn.send(("done", rank, jid))
            else:
                raise RuntimeError(f"unknown message {msg[0]!r}")
        except BaseException as exc:       # noqa: BLE001 - outside a guarded chunk: this child is no longer trustworthy
            try:
                conn.send(("error", rank, f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"))
            except (OSError, BrokenPipeError):
                pass
            break
    drop_job()
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
        close = getattr(engine, "close", None)
        if close:
            close()
    except BaseException:                  # noqa: BLE001
        pass


# ------------------------------------------------------------------------------------------------ parent
_DEAL = os.environ.get("MANGA_OCR_DEAL", "equal")                     # "equal" | "guided"


def deal_sizes(n: int, world: int, max_chunk: int, min_chunk: int = _MIN_CHUNK, policy: Optional[str] = None) -> List[Tuple[int, int]]:
    """The queue [0, n) cut into the chunks that are dealt, in order (the parent of `MultiGpuEngine` deals them to whichever
    child has room; `bench.py --queue` lets its SPMD ranks pull them from a shared counter).

    "equal" (default): whole rounds of EQUAL fat chunks - ceil(n / (world x max_chunk)) rounds of `world` chunks, each at
    most what one child's engine decodes at once (`max_chunk`).  A decode's cost is mostly its step count, not its rows
    (one GPU, 300 steps: 64 rows 50 ms, 256 rows 91 ms, 1250 rows 231 ms, 2560 rows 305 ms), so on equal workers the
    fattest chunks win: 10,000 crops on 8 children are 8 chunks of 1250 - the static shards, dealt; a child that is
    late, slow or dead still only costs what it holds (re-dealt on failure).
    "guided" (r03's schedule; MANGA_OCR_DEAL=guided): a chunk is the rows left divided by twice the workers, at most
    `max_chunk`, at least `min_chunk` - fat chunks first, small ones last: children of UNEQUAL speed finish together, at
    the price of a tail of small decodes (625, 586, 549 ... 64 rows for the queue above: ~3 x the time on a healthy node)."""
    policy = policy or _DEAL
    if policy not in ("equal", "guided"):
        raise ValueError(f"MANGA_OCR_DEAL / policy must be 'equal' or 'guided', not {policy!r}")
    world = max(world, 1)
    out, lo = [], 0
    if policy == "equal":
        if n <= 0:
            return out
        rounds = -(-n // (world * max_chunk))
        k = min(rounds * world, -(-n // max(1, min(min_chunk, max_chunk))))      # never thinner than min_chunk (but cover n)
        k = max(k, -(-n // max_chunk))
        base, extra = divmod(n, k)
        for i in range(k):
            hi = lo + base + (1 if i < extra else 0)
            out.append((lo, hi))
            lo = hi
        return out
    while lo < n:
        c = max(min_chunk, min(max_chunk, -(-(n - lo) // (2 * world))))
        hi = min(n, lo + c)
        out.append((lo, hi))
        lo = hi
    return out


class MultiGpuEngine:
    """The parent-side handle: same ``recognize_images`` / ``recognize_regions`` surface as ``Engine``; thread-safe."""

    def __init__(self, devices: Sequence[int], factory: Callable = default_engine_factory, factory_args: Optional[dict] = None,
                 backend: str = "nccl", max_chunk: Optional[int] = None, min_chunk: int = _MIN_CHUNK):
        import multiprocessing as mp
        if len(devices) < 1:
            raise ValueError("devices must name at least one GPU")
        self.devices = [int(d) for d in devices]
        self.world = len(self.devices)
        fa = dict(factory_args or {})
        # rows one child decodes at once: every lane of its engine gets a full internal batch
        self.max_chunk = int(max_chunk or fa.get("max_batch", 1024) * max(1, fa.get("lanes", 2)))
        self.min_chunk = max(1, min(int(min_chunk), self.max_chunk))
        self._lock = threading.Lock()          # one job at a time on the pipes
        self._broken: Optional[str] = None
        self._closed = False
        self._job_seq = 0
        self.stats: Dict[str, object] = {}     # of the last job: chunks dealt per child, re-deals, lost children
        ctx = mp.get_context("spawn")          # fresh interpreters: no GPU state is inherited, nothing is re-exec'ed
        port = _free_port()
        self._conns, self._procs = [], []
        for r, d in enumerate(self.devices):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker_main, args=(r, self.world, d, port, backend, factory, fa, child),
                            name=f"mocr-gpu{d}", daemon=True)
            p.start()
            child.close()
            self._conns.append(parent)
            self._procs.append(p)
        self._alive = [True] * self.world
        self.max_len = None
        try:
            for r, c in enumerate(self._conns):
                if not c.poll(_TIMEOUT):
                    raise RuntimeError(f"GPU worker {r} (device {self.devices[r]}) did not come up within {_TIMEOUT:.0f} s")
                try:
                    msg = c.recv()
                except EOFError:
                    raise RuntimeError(f"GPU worker {r} (device {self.devices[r]}) died while starting") from None
                if msg[0] == "error":
                    raise RuntimeError(f"GPU worker {msg[1]} failed: {msg[2]}")
                self.max_len = int(msg[2])
        except BaseException:
            self._teardown()
            raise

    # ------------------------------------------------------------------ plumbing
    def _teardown(self) -> None:
        """Terminate every child this object started (the exact processes, never a pattern)."""
        self._closed = True
        for c in self._conns:
            try:
                c.send(("stop",))
            except (OSError, BrokenPipeError, ValueError):
                pass
        deadline = time.monotonic() + 10.0
        for p in self._procs:
            p.join(timeout=max(0.1, deadline - time.monotonic()))
        for p in self._procs:
            if p.is_alive():
                p.terminate()
        for p in self._procs:
            p.join(timeout=5)
        for c in self._conns:
            try:
                c.close()
            except OSError:
                pass

    def _fatal(self, why: str):
        self._broken = why
        self._teardown()
        raise RuntimeError(f"multi-GPU dispatcher is broken: {why}")

    def _run(self, n: int, payload_bytes: int, fill: Callable[[np.ndarray], None], header: dict):
        with self._lock:
            if self._broken:
                raise RuntimeError(f"multi-GPU dispatcher is broken: {self._broken} (build a new MultiGpuEngine)")
            if self._closed:
                raise RuntimeError("multi-GPU dispatcher is closed")
            return self._run_locked(n, payload_bytes, fill, header)

    def _run_locked(self, n, payload_bytes, fill, header):
        L = self.max_len
        self._job_seq += 1
        jid = self._job_seq
        shm_in = shared_memory.SharedMemory(create=True, size=max(payload_bytes, 1))
        shm_out = shared_memory.SharedMemory(create=True, size=max(n * (L + 1) * 4, 4))
        try:
            buf = np.ndarray((shm_in.size,), dtype=np.uint8, buffer=shm_in.buf)
            fill(buf)
            del buf
            o = np.ndarray((n, L + 1), dtype=np.int32, buffer=shm_out.buf)
            o[:, :L] = 0
            o[:, L] = -1                                     # a row nobody delivers reads "failed"
            del o
            head = dict(header, n=n, shm_in=shm_in.name, shm_out=shm_out.name)
            live = [r for r in range(self.world) if self._alive[r]]
            for r in live:
                self._send(r, ("job", jid, head))
            # ---- deal: (lo, hi, attempts, avoid-rank, deaths)
            queue = deque((lo, hi, 0, -1, 0) for lo, hi in deal_sizes(n, len(live), self.max_chunk, self.min_chunk))
            out: Dict[int, List[tuple]] = {r: [] for r in live}      # chunks a child holds, undecided
            kept: Dict[int, List[tuple]] = {r: [] for r in live}     # chunks a child has decoded and keeps until the exchange
            streak = {r: 0 for r in live}                            # consecutive failed chunks of a child
            failed: Dict[int, str] = {}
            stats = {"chunks": {r: 0 for r in live}, "redealt": 0, "lost": []}

            def lose(r, why):
                """Drop child r: what it was decoding AND what it had decoded go back to the queue for the survivors."""
                if not self._alive[r]:
                    return
                self._alive[r] = False
                stats["lost"].append((r, why))
                for (lo, hi, att, _av, deaths) in out.pop(r, []):
                    if deaths >= 1:        # this chunk has now been on two children that died: do not deal it a third time
                        for row in range(lo, hi):
                            failed[row] = f"two workers died decoding the chunk [{lo}, {hi}) ({why})"
                    else:
                        queue.appendleft((lo, hi, att, r, deaths + 1))
                for (lo, hi) in kept.pop(r, []):
                    queue.append((lo, hi, 0, r, 0))
                streak.pop(r, None)
                try:
                    self._procs[r].terminate()       # the exact child this object started
                except Exception:      # noqa: BLE001
                    pass

            def deal():
                # breadth first: every child gets its first chunk before anyone gets a second (eight equal chunks on eight
                # children are one each)
                for depth in range(1, _DEPTH + 1):
                    for r in list(out):
                        while r in out and len(out[r]) < depth and queue:
                            # the first chunk this child is allowed to take (a re-dealt chunk avoids the child it failed
                            # on, unless that is the only one left)
                            pick = next((i for i, c in enumerate(queue) if c[3] != r or len(out) == 1), None)
                            if pick is None:
                                break
                            c = queue[pick]
                            del queue[pick]
                            try:
                                self._conns[r].send(("chunk", jid, c[0], c[1]))
                            except (OSError, BrokenPipeError, ValueError):
                                queue.appendleft(c)
                                lose(r, "pipe closed")
                                break
                            out[r].append(c)
                            stats["chunks"][r] += 1

            deal()
            while any(out.values()) or queue:
                if not out:
                    self._fatal("every GPU worker was lost")
                if not any(out.values()):
                    deal()
                    if not any(out.values()):      # chunks are left but nobody may take them
                        self._fatal("no worker can take the remaining chunks")
                    continue
                ready = mp_connection.wait([self._conns[r] for r in out if out[r]], timeout=_TIMEOUT)
                if not ready:
                    for r in [r for r in out if out[r]]:
                        lose(r, f"no answer within {_TIMEOUT:.0f} s")
                    deal()
                    continue
                for c in ready:
                    r = self._conns.index(c)
                    try:
                        msg = c.recv()
                    except (EOFError, OSError):
                        lose(r, "process died")
                        continue
                    if msg[0] == "error":
                        lose(r, msg[2].splitlines()[0])
                        continue
                    _, _rk, mjid, lo, hi, err = msg
                    ent = next((e for e in out[r] if e[0] == lo and e[1] == hi), None)
                    if mjid != jid or ent is None:
                        self._fatal(f"worker {r} answered for a chunk it was not dealt")
                    out[r].remove(ent)
                    if err is None:
                        kept[r].append((lo, hi))
                        streak[r] = 0
                        continue
                    stats["redealt"] += 1
                    streak[r] += 1
                    _lo, _hi, att, _av, deaths = ent
                    if att == 0 and len(out) > 1:              # once more, on another child
                        queue.appendleft((lo, hi, 1, r, deaths))
                    elif hi - lo > 1:                          # still failing: halve it until the bad crop stands alone
                        mid = (lo + hi) // 2
                        queue.appendleft((mid, hi, 1, -1, deaths))
                        queue.appendleft((lo, mid, 1, -1, deaths))
                    else:
                        failed[lo] = f"worker {r}: {err}"
                    if streak[r] >= 4 and len(out) > 1 and any(kept[q] or out[q] for q in out if q != r):
                        lose(r, f"four chunks in a row failed, last: {err}")     # e.g. a poisoned engine: stop feeding it
                deal()
            # ---- the exchange step
            live = [r for r in range(self.world) if self._alive[r]]
            mode = "allgather" if len(live) == self.world else "direct"
            n_max = max([sum(hi - lo for lo, hi in kept.get(r, [])) for r in live] + [1])
            for r in live:
                self._send(r, ("gather", jid, mode, n_max))
            for r in live:
                c = self._conns[r]
                if not c.poll(_TIMEOUT):
                    self._fatal(f"worker {r} (device {self.devices[r]}) did not finish the exchange step within {_TIMEOUT:.0f} s")
                try:
                    msg = c.recv()
                except (EOFError, OSError):
                    self._fatal(f"worker {r} (device {self.devices[r]}) died in the exchange step")
                if msg[0] != "done" or msg[2] != jid:
                    self._fatal(f"worker {r} failed in the exchange step: {msg[2] if msg[0] == 'error' else msg!r}")
            stats["mode"] = mode
            self.stats = stats
            o = np.ndarray((n, L + 1), dtype=np.int32, buffer=shm_out.buf)
            ids, lens = o[:, :L].copy(), o[:, L].copy()
            del o
            missing = [int(i) for i in np.nonzero(lens < 0)[0] if int(i) not in failed]
            for i in missing:
                failed[i] = "row was never delivered"
            if failed:
                raise ShardError(failed, ids, lens)
            return ids, lens
        finally:
            for s in (shm_in, shm_out):
                s.close()
                s.unlink()

    def _send(self, r, msg):
        try:
            self._conns[r].send(msg)
        except (OSError, BrokenPipeError, ValueError):
            self._fatal(f"worker {r} (device {self.devices[r]}) is gone")

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
    def recognize_images(self, images, bgr: bool = False, rotate=None) -> Tuple[np.ndarray, np.ndarray]:
        n = len(images)
        if n == 0:
            return np.zeros((0, self.max_len), np.int32), np.zeros(0, np.int32)
        if rotate is not None:
            # checked HERE: a short or out-of-range list would otherwise fail inside a child, chunk by chunk, and come back as
            # obscure per-row errors after many re-dealt decodes
            rotate = [int(r) for r in rotate]
            if len(rotate) != n or any(r not in (0, 1, 2) for r in rotate):
                raise ValueError(f"recognize_images: rotate needs one code in {{0, 1, 2}} per image ({n} images, {len(rotate)} codes)")
        descs, size, fill = self._pack(images)
        return self._run(n, size, fill, dict(kind="images", descs=descs, bgr=bool(bgr), rotate=rotate))

    def recognize_regions(self, pages, regions, bgr: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        regs = [tuple(int(v) for v in r) for r in regions]
        n = len(regs)
        if n == 0:
            return np.zeros((0, self.max_len), np.int32), np.zeros(0, np.int32)
        pdescs, size, fill = self._pack(pages)
        return self._run(n, size, fill, dict(kind="regions", descs=pdescs, regs=regs, bgr=bool(bgr)))

    @property
    def alive(self) -> List[int]:
        return [r for r in range(self.world) if self._alive[r]]

    def close(self) -> None:
        with self._lock:
            if self._closed:
                return
            self._teardown()

    def __del__(self):
        try:
            if not getattr(self, "_closed", True):
                self._teardown()
        except Exception:
            pass
