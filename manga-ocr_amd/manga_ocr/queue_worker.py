"""Crop-job queue in front of the recogniser: the micro-batching counterpart of the reference's worker pool.

The reference drains its job list with up to 15 ``QueueProcessorWorker`` threads, each doing ONE crop at a time
(``src/core/workers.py:209-247``; queue and pool in ``src/ui/main_window.py:4286-4335``).  Here one worker takes up to
``max_batch`` jobs per round and hands the recogniser all their crops at once, keeping the reference's semantics:

* FIFO order, completions reported in submission order;
* per-job error isolation (``workers.py:241-244``: an exception costs that job only - a failing batch is retried
  crop by crop so that the other jobs of the batch still complete);
* what happens to a crop before and after the recogniser on the Manga-OCR branch (``workers.py:318-327`` and
  ``main_window.py:9774-9803``): the orientation-only rotation, BGR -> RGB, then the whitespace join and the
  failure-sentinel test (``main_window.py:3808-3809``, ``workers.py:296, 356``);
* the crop a detected text region gets in the Text-detect path: bounding box padded by 8 % of its longer side and
  clipped to the page (``main_window.py:9530-9540``).

Host-side plumbing only (numpy); translation, typesetting and the Qt signals stay in the application.
"""
from __future__ import annotations

import threading
from dataclasses import dataclass, field
from typing import Any, Callable, List, Optional, Sequence, Tuple

import numpy as np

ERROR_SENTINELS = ("[ERROR:", "[TESSERACT ERROR:")


def orient_crop(bgr: np.ndarray, orientation: str = "Auto-Detect") -> np.ndarray:
    """The only change the Manga-OCR branch makes to a crop: "Vertical" with a landscape crop -> 90 degrees clockwise,
    "Horizontal" with a portrait crop -> 90 degrees counter-clockwise, anything else untouched."""
    h, w = bgr.shape[:2]
    if orientation == "Vertical" and w > h:
        return np.ascontiguousarray(np.rot90(bgr, k=-1))      # cv2.ROTATE_90_CLOCKWISE
    if orientation == "Horizontal" and h > w:
        return np.ascontiguousarray(np.rot90(bgr, k=1))       # cv2.ROTATE_90_COUNTERCLOCKWISE
    return bgr


ROTATE_NONE, ROTATE_90_CW, ROTATE_90_CCW = 0, 1, 2        # include/mocr.h: MOCR_ROTATE_*


def rotation_code(h: int, w: int, orientation: str = "Auto-Detect") -> int:
    """The same rule as :func:`orient_crop`, as the code the engine's resize takes (``mocr_image.rotate``): the rotation
    then happens in the device's source addressing and the crop is handed over as it lies in memory."""
    if orientation == "Vertical" and w > h:
        return ROTATE_90_CW
    if orientation == "Horizontal" and h > w:
        return ROTATE_90_CCW
    return ROTATE_NONE


def bgr_to_rgb(bgr: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(bgr[..., ::-1])


def padded_region_crop(page: np.ndarray, x: int, y: int, w: int, h: int) -> Optional[np.ndarray]:
    """Crop of a detected region: the box grown by int(0.08 * max(w, h)) on every side, clipped to the page; None when
    nothing more than a 1-pixel sliver is left."""
    ph, pw = page.shape[:2]
    pad = int(max(w, h) * 0.08)
    x1, y1 = max(x - pad, 0), max(y - pad, 0)
    x2, y2 = min(x + w + pad, pw), min(y + h + pad, ph)
    if x2 - x1 <= 1 or y2 - y1 <= 1:
        return None
    return page[y1:y2, x1:x2].copy()


def clean_and_join(text: str) -> str:
    return " ".join(text.split())


def ocr_failed(raw_text: str) -> bool:
    """The caller's failure test: nothing left after the whitespace join, or an engine error marker in the text."""
    return (not clean_and_join(raw_text)) or any(s in raw_text for s in ERROR_SENTINELS)


@dataclass
class CropJob:
    crop_bgr: np.ndarray                      # uint8 [h,w,3] BGR as produced by the crop tools
    orientation: str = "Auto-Detect"
    payload: Any = None                       # whatever the application needs back (page path, rect, settings ...)
    pre_detected_text: Optional[str] = None   # Text-detect mode: OCR already done, skip the recogniser
    seq: int = field(default=-1, compare=False)


class CropJobQueue:
    """FIFO job list + one micro-batching worker thread.

    ``recognize(list_of_rgb_uint8_arrays) -> list[str]`` is e.g. ``lambda crops: reader.recognize_batch(map(Image.fromarray, crops))``
    or a function over ``Engine.recognize_images`` + tokenizer decode.  ``on_complete(job, raw_text)`` and ``on_error(job, exc)``
    are called from the worker thread, in submission order."""

    def __init__(self, recognize: Optional[Callable[[List[np.ndarray]], Sequence[str]]], on_complete: Callable[[CropJob, str], None],
                 on_error: Optional[Callable[[CropJob, BaseException], None]] = None, max_batch: int = 64,
                 recognize_oriented: Optional[Callable[[List[np.ndarray], List[str]], Sequence[str]]] = None):
        """``recognize_oriented(bgr_crops, orientations) -> texts`` (e.g. ``MangaOcr.recognize_bgr``) takes the jobs' crops
        exactly as the crop tools produced them - BGR, unrotated - together with their orientation settings: rotation
        and channel order are then the engine's business (device side).  Without it the worker rotates and swaps the
        channels on the host (the reference's own order of operations) and calls ``recognize(rgb_crops)``."""
        if recognize is None and recognize_oriented is None:
            raise ValueError("give recognize or recognize_oriented")
        self._recognize, self._done, self._err = recognize, on_complete, on_error or (lambda job, exc: None)
        self._oriented = recognize_oriented
        self.max_batch = max_batch
        self._jobs: List[CropJob] = []
        self._cv = threading.Condition()
        self._seq = 0
        self._stop = False
        self._idle = True
        self._thread = threading.Thread(target=self._run, name="mocr-queue", daemon=True)
        self._thread.start()

    def submit(self, job: CropJob) -> int:
        with self._cv:
            if self._stop:
                raise RuntimeError("queue is closed")
            job.seq = self._seq
            self._seq += 1
            self._jobs.append(job)
            self._cv.notify_all()
            return job.seq

    def pending(self) -> int:
        with self._cv:
            return len(self._jobs)

    def join(self, timeout: Optional[float] = None) -> bool:
        """Block until every submitted job has been reported."""
        with self._cv:
            return self._cv.wait_for(lambda: not self._jobs and self._idle, timeout)

    def close(self) -> None:
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._thread.join(timeout=10)

    # ------------------------------------------------------------------ worker
    def _take(self) -> List[CropJob]:
        with self._cv:
            while not self._jobs and not self._stop:
                self._idle = True
                self._cv.notify_all()
                self._cv.wait()
            self._idle = False
            batch, self._jobs = self._jobs[:self.max_batch], self._jobs[self.max_batch:]
            return batch

    def _run(self) -> None:
        while True:
            batch = self._take()
            if not batch:
                with self._cv:
                    self._idle = True
                    self._cv.notify_all()
                return
            results: List[Tuple[CropJob, Optional[str], Optional[BaseException]]] = []
            todo = [j for j in batch if not j.pre_detected_text]
            texts: dict = {}
            errors: dict = {}
            def run(jobs):
                if self._oriented is not None:
                    return list(self._oriented([j.crop_bgr for j in jobs], [j.orientation for j in jobs]))
                return list(self._recognize([bgr_to_rgb(orient_crop(j.crop_bgr, j.orientation)) for j in jobs]))

            if todo:
                try:
                    out = run(todo)
                    if len(out) != len(todo):
                        raise RuntimeError("recogniser returned a different number of texts than crops")
                    texts = {j.seq: t for j, t in zip(todo, out)}
                except BaseException:
                    # one bad crop must not cost its neighbours their result: redo the batch crop by crop
                    for j in todo:
                        try:
                            texts[j.seq] = run([j])[0]
                        except BaseException as exc:      # noqa: BLE001 - reported per job, the loop lives on
                            errors[j.seq] = exc
            for j in batch:
                if j.pre_detected_text:
                    results.append((j, j.pre_detected_text, None))
                elif j.seq in errors:
                    results.append((j, None, errors[j.seq]))
                else:
                    results.append((j, texts[j.seq], None))
            for j, text, exc in results:                   # submission order
                try:
                    if exc is not None:
                        self._err(j, exc)
                    else:
                        self._done(j, text)
                except BaseException:                      # noqa: BLE001 - a callback's own failure is the application's business
                    pass
