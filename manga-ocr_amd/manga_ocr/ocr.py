"""``MangaOcr`` - the drop-in for the recogniser the reference application constructs once
(``src/ui/main_window.py:3394``: ``MangaOcr()``) and calls per crop from many threads
(``src/ui/main_window.py:9801``: ``self.manga_ocr_reader(pil_img) -> str``).

Same call surface as the ``manga-ocr`` package [RECALL]: ``MangaOcr(pretrained_model_name_or_path=
'kha-white/manga-ocr-base', force_cpu=False)``, ``__call__(PIL.Image | str | Path) -> str`` and
``ValueError`` for anything else.  Behind it: the HIP engine (libmocr_hip.so) on this
process's MI355X.  Concurrent callers (the reference runs up to 15 QueueProcessorWorker threads
on one shared instance, ``src/core/workers.py:209-247``) are coalesced into one batch per decode
instead of being serialised.  There is no CPU path: without the library or a GPU the constructor
raises, which the application already handles (``main_window.py:3396-3398``).
"""
from __future__ import annotations

import glob
import os
import threading
import time
from concurrent.futures import Future
from pathlib import Path
from typing import List, Optional, Sequence, Union

import numpy as np

from .engine import Engine
from .text import Vocab, find_vocab, ids_to_text
from .weights import DEFAULT_SPEC, load_checkpoint, synthetic_weights

DEFAULT_MODEL = "kha-white/manga-ocr-base"


def _resolve_model_dir(name_or_path: str) -> Optional[str]:
    """A local directory, $MANGA_OCR_MODEL_DIR, or an already-downloaded HF cache snapshot.
    Never touches the network."""
    cands = [name_or_path, os.environ.get("MANGA_OCR_MODEL_DIR", "")]
    hub = os.environ.get("HF_HOME", os.path.join(os.path.expanduser("~"), ".cache", "huggingface"))
    cands += sorted(glob.glob(os.path.join(hub, "hub", "models--" + name_or_path.replace("/", "--"), "snapshots", "*")))
    for c in cands:
        if c and os.path.isdir(c) and os.path.exists(os.path.join(c, "config.json")):
            return c
    return None


def to_gray224(img, size: int = 224) -> np.ndarray:
    """PIL image -> uint8 [224,224] ON THE HOST: ``convert('L')`` then the HF image processor's
    ``resize((224,224), BILINEAR)``.  Kept for callers that want the plane itself; the recogniser's own
    path (:func:`to_pixels`) leaves both steps to the device, which is bit-exact with this."""
    from PIL import Image
    g = img.convert("L")
    if g.size != (size, size):
        g = g.resize((size, size), Image.BILINEAR)
    return np.asarray(g, dtype=np.uint8)


def to_pixels(img) -> np.ndarray:
    """PIL image -> the uint8 array handed to the engine: [h,w,3] for RGB crops (what the reference builds at
    ``src/ui/main_window.py:9796-9800``), [h,w] for L; any other mode goes through Pillow's own convert('L')
    first (palettes, alpha, 16-bit ...: rare, and not worth restating).  ``convert('L')`` of RGB and the
    BILINEAR resize to 224x224 then run on the device (csrc/preprocess.h)."""
    if img.mode not in ("RGB", "L"):
        img = img.convert("L")
    return np.asarray(img, dtype=np.uint8)


class _Batcher:
    """Coalesces concurrent single-crop requests into engine batches (FIFO, per-request error
    isolation like the reference's worker loop, ``src/core/workers.py:241-244``)."""

    def __init__(self, engine: Engine, max_batch: int, timeout_ms: float):
        self.engine, self.max_batch, self.timeout = engine, max_batch, timeout_ms / 1000.0
        self._q: List = []
        self._cv = threading.Condition()
        self._stop = False
        self._thread = threading.Thread(target=self._run, name="mocr-batcher", daemon=True)
        self._thread.start()

    def submit(self, gray: np.ndarray) -> Future:
        f: Future = Future()
        with self._cv:
            if self._stop:
                raise RuntimeError("MangaOcr is closed")
            self._q.append((gray, f))
            self._cv.notify()
        return f

    def _run(self):
        while True:
            with self._cv:
                while not self._q and not self._stop:
                    self._cv.wait()
                if self._stop and not self._q:
                    return
                deadline = time.monotonic() + self.timeout
                while len(self._q) < self.max_batch and not self._stop:
                    left = deadline - time.monotonic()
                    if left <= 0:
                        break
                    self._cv.wait(left)
                batch, self._q = self._q[:self.max_batch], self._q[self.max_batch:]
            try:
                ids, lens = self.engine.recognize_images([g for g, _ in batch])
                for i, (_, f) in enumerate(batch):
                    f.set_result(ids[i, :lens[i]].copy())
            except BaseException as exc:  # every waiting caller gets the error; the loop lives on
                for _, f in batch:
                    if not f.done():
                        f.set_exception(exc)

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._thread.join(timeout=5)


# HBM one row of a lane's workspace takes (bf16): 197 encoder rows x (fp32 residual + LN out + QKV + context + FFN
# intermediate + encoder output = 18,432 B) + latent key/value rows + input planes
_BYTES_PER_ROW_PER_LANE = 197 * 18432 + 2 * 300 * 768 * 2 + 4 * 224 * 224


def default_max_batch(device: int, lanes: int) -> int:
    """Rows per internal batch when the caller does not say: as fat as a fifth of the free HBM allows, capped at
    2048 (decode throughput is bought with fat batches - DESIGN.md §5 - and flattens beyond that), at least 64."""
    from .engine import device_memory
    free, _total = device_memory(device)
    rows = int(free * 0.2 / (max(1, lanes) * _BYTES_PER_ROW_PER_LANE))
    return max(64, min(2048, rows // 64 * 64))


class MangaOcr:
    def __init__(self, pretrained_model_name_or_path: str = DEFAULT_MODEL, force_cpu: bool = False, *,
                 dtype: Optional[str] = None, device: Optional[int] = None, devices: Optional[Sequence[int]] = None,
                 max_batch: Optional[int] = None, lanes: Optional[int] = None, batch_timeout_ms: Optional[float] = None,
                 synthetic_seed: Optional[int] = None):
        """``MangaOcr()`` as the application calls it (``src/ui/main_window.py:3394``) builds the engine on this
        process's GPU with two lanes and an internal batch sized from the free HBM.  ``devices=[0, 1, ...]`` (or
        ``MANGA_OCR_DEVICES=0,1,...``) instead starts one child process per GPU and shards every batch call over
        them (``manga_ocr/multi.py``); the parent then never touches a GPU."""
        if force_cpu:
            raise RuntimeError("this MangaOcr is the MI355X engine: there is no CPU path (force_cpu=True is not supported)")
        dtype = dtype or os.environ.get("MANGA_OCR_DTYPE", "bf16")
        if devices is None and os.environ.get("MANGA_OCR_DEVICES"):
            devices = [int(x) for x in os.environ["MANGA_OCR_DEVICES"].split(",") if x.strip() != ""]
        device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
        lanes = int(lanes or os.environ.get("MANGA_OCR_LANES", "2"))
        if max_batch is None and os.environ.get("MANGA_OCR_MAX_BATCH"):
            max_batch = int(os.environ["MANGA_OCR_MAX_BATCH"])
        if synthetic_seed is None and os.environ.get("MANGA_OCR_SYNTHETIC"):
            synthetic_seed = int(os.environ["MANGA_OCR_SYNTHETIC"])
        model_dir = None
        if synthetic_seed is not None:
            spec, weights, vocab = DEFAULT_SPEC, None, Vocab.synthetic(DEFAULT_SPEC.vocab)
        else:
            model_dir = _resolve_model_dir(str(pretrained_model_name_or_path))
            if model_dir is None:
                raise FileNotFoundError(
                    f"no local copy of '{pretrained_model_name_or_path}' (looked at the path, $MANGA_OCR_MODEL_DIR and the "
                    "HF cache; this build never downloads). Set MANGA_OCR_SYNTHETIC=<seed> for synthetic weights.")
            spec, weights = load_checkpoint(model_dir)
            vp = find_vocab(model_dir)
            if vp is None:
                raise FileNotFoundError(f"vocab.txt not found in {model_dir}")
            vocab = Vocab.from_file(vp)
            if len(vocab) != spec.vocab:
                raise ValueError(f"vocab.txt has {len(vocab)} entries, config says {spec.vocab}")
        self.spec, self.vocab = spec, vocab
        # what the checkpoint's config.json asked of generate() and this engine ignores (greedy decode only): the reference
        # application's recogniser would honour e.g. num_beams=4 / no_repeat_ngram_size=3, so on such a checkpoint the
        # strings can differ from the pip package's; {} when there is nothing to report
        self.ignored_generation_config = dict(getattr(spec, "ignored_generation", ()))
        if devices is not None and len(devices) > 1:
            from .multi import MultiGpuEngine
            max_batch = int(max_batch or 2048)
            self.engine = MultiGpuEngine(devices, factory_args=dict(synthetic_seed=synthetic_seed, model_dir=model_dir, dtype=dtype,
                                                                    max_batch=max_batch, lanes=lanes))
        else:
            if devices:
                device = int(devices[0])
            if weights is None:
                weights = synthetic_weights(synthetic_seed)
            max_batch = int(max_batch or default_max_batch(device, lanes))
            self.engine = Engine(weights, spec, dtype=dtype, device=device, max_batch=max_batch, lanes=lanes)
        self.max_batch = max_batch
        # How long the first single-crop caller waits for company before its batch is sent off.  Callers that arrive while a
        # batch is being decoded queue up behind it and form the next batch anyway, so the window only has to catch a burst
        # of workers that start together.  MI355X, 24-token texts (tools/call_latency.py): 2.0 / 0.3 / 0 ms -> one caller
        # 5.3 / 3.6 / 3.2 ms per call, 15 worker threads 2100 / 2350 / 1880 crops/s
        if batch_timeout_ms is None:
            batch_timeout_ms = float(os.environ.get("MANGA_OCR_BATCH_WINDOW_MS", "0.3"))
        self._batcher = _Batcher(self.engine, max_batch, batch_timeout_ms)
        # same warm-up the reference's recogniser does in its constructor (one inference)
        self.recognize_ids([np.zeros((spec.image_size, spec.image_size), dtype=np.uint8)])

    # ------------------------------------------------------------------ reference call surface
    def __call__(self, img_or_path) -> str:
        from PIL import Image
        if isinstance(img_or_path, (str, Path)):
            img = Image.open(img_or_path)
        elif isinstance(img_or_path, Image.Image):
            img = img_or_path
        else:
            raise ValueError(f"img_or_path must be a path or PIL.Image, instead got: {img_or_path}")
        ids = self._batcher.submit(to_pixels(img)).result()
        return ids_to_text(self.vocab, ids)

    # ------------------------------------------------------------------ batch surface (callers that hold many crops)
    def recognize_ids(self, crops: Sequence[np.ndarray], bgr: bool = False, rotate: Optional[Sequence[int]] = None) -> List[np.ndarray]:
        """uint8 crops of any sizes ([h,w] luminance or [h,w,3] RGB; BGR with ``bgr=True``; ``rotate``: per crop 0 / 1 (90
        degrees clockwise) / 2 (counter-clockwise), done by the device) -> token ids (without padding)."""
        ids, lens = self.engine.recognize_images(list(crops), bgr, rotate)
        return [ids[i, :lens[i]].copy() for i in range(len(lens))]

    def recognize_batch(self, images: Sequence) -> List[str]:
        """All crops of a page (or chapter) at once - what ``_collect_manga_detections``
        (``src/ui/main_window.py:9462-9476``) does one region at a time."""
        return [ids_to_text(self.vocab, r) for r in self.recognize_ids([to_pixels(im) for im in images])]

    def recognize_batch_arrays(self, crops: Sequence[np.ndarray]) -> List[str]:
        """uint8 arrays ([h,w] luminance or [h,w,3] RGB, any sizes) -> strings: what a caller that already holds numpy
        crops (the crop-job queue) uses instead of wrapping each one in a PIL image."""
        return [ids_to_text(self.vocab, r) for r in self.recognize_ids(list(crops))]

    def recognize_bgr(self, crops_bgr: Sequence[np.ndarray], orientations: Optional[Sequence[str]] = None) -> List[str]:
        """BGR crops exactly as the crop tools and the worker hold them (``cropped_cv_img``, ``src/core/workers.py:300``):
        the BGR -> RGB swap of ``src/ui/main_window.py:9800`` is folded into the device's luminance conversion, and with
        ``orientations`` (the jobs' "Auto-Detect" / "Vertical" / "Horizontal" settings) the orientation-only rotation of
        ``src/core/workers.py:320-326`` / ``src/ui/main_window.py:9787-9795`` into the device's resize addressing."""
        from .queue_worker import rotation_code
        crops = list(crops_bgr)
        rot = None
        if orientations is not None:
            if len(orientations) != len(crops):        # zip() would truncate silently: a short list must not cost a decode
                raise ValueError(f"recognize_bgr: {len(crops)} crops but {len(orientations)} orientations")
            rot = [rotation_code(c.shape[0], c.shape[1], o) for c, o in zip(crops, orientations)]
        return [ids_to_text(self.vocab, r) for r in self.recognize_ids(crops, bgr=True, rotate=rot)]

    def recognize_regions(self, pages_bgr: Sequence[np.ndarray], regions) -> List[str]:
        """``regions``: (page_index, x, y, w, h) bounding rectangles on BGR pages; every page is uploaded once and
        the padded crops (``src/ui/main_window.py:9530-9540``) are cut on the device.  One string per region
        ('' for a region reduced to a sliver, like the reference)."""
        ids, lens = self.engine.recognize_regions(list(pages_bgr), list(regions), True)
        return [ids_to_text(self.vocab, ids[i, :lens[i]]) if lens[i] > 0 else "" for i in range(len(lens))]

    def recognize_page(self, page_bgr: np.ndarray, regions):
        """``_collect_manga_detections`` for one page: ``regions`` = the detector's (text, polygon) pairs."""
        from .regions import recognize_page
        return recognize_page(self, page_bgr, regions)

    def recognize_pages(self, pages_bgr, regions_per_page, on_error=None):
        """``AutoDetectorWorker.run`` (Text mode) over several pages: one job queue for all regions of all pages."""
        from .regions import recognize_pages
        return recognize_pages(self, pages_bgr, regions_per_page, on_error)

    def close(self) -> None:
        self._batcher.close()
        self.engine.close()
