"""Crop-job producers and the page-level Text-detect driver (§8(f) row 2).

Host-side integer work on a handful of polygons per page (numpy); the pixels themselves go to the device once per
page (``Engine.recognize_regions`` -> ``mocr_recognize_regions``), where each region's crop is cut by the resize
kernel's descriptor.

What the reference does, restated from its call sites:

* ``process_rect_area`` (``src/ui/main_window.py:6399-6447``) and ``process_confirmed_polygon`` (``:6481-6527``) turn a
  selection into a crop job: ``page.crop((x, y, rect.right(), rect.bottom()))`` - Qt's ``right() = x + w - 1`` and
  PIL's exclusive box make the crop (w-1) x (h-1) - converted RGB -> BGR; the polygon variant paints everything
  outside the polygon white (``:6499-6506``: ``cv2.fillPoly`` mask, ``bitwise_and`` / ``add``).
* ``_collect_manga_detections`` / ``_recognize_polygon`` (``:9462-9476, 9530-9549``): per detected region the bounding
  box padded by 8 % of its longer side and clipped to the page goes through ``perform_ocr`` (orientation
  'Auto-Detect' = no rotation), the result is ``.strip()``-ed, and ``recognized or text`` is kept.
* ``AutoDetectorWorker.run`` (``src/core/workers.py:448-482``, Text mode): per file ``[{'polygon', 'text'}]``; a failing
  page reports an error and the loop continues.

``cv2`` is not installed in the build image, so ``fill_poly_mask`` restates OpenCV's ``fillPoly`` for ``LINE_8``,
``shift = 0`` [RECALL of modules/imgproc/src/drawing.cpp, 4.x: ``CollectPolyEdges`` draws every edge with the
8-connected ``Line`` (after ``clipLine``) and ``FillEdgeCollection`` fills the even-odd interior spans walking the
edges in 16.16 fixed point]; it is pinned by hand-derived fixtures in ``tests/test_regions.py``, not by OpenCV itself.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT

Point = Tuple[int, int]


def bounding_rect(points: Sequence[Point]) -> Tuple[int, int, int, int]:
    """``QPolygon.boundingRect()``: (x, y, w, h) with w = max_x - min_x + 1 (Qt's inclusive integer rectangle);
    an empty polygon gives (0, 0, 0, 0)."""
    if len(points) == 0:
        return 0, 0, 0, 0
    xs = [int(p[0]) for p in points]
    ys = [int(p[1]) for p in points]
    return min(xs), min(ys), max(xs) - min(xs) + 1, max(ys) - min(ys) + 1


def _trunc_div(a: int, b: int) -> int:
    """C++ integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _clip_line(w: int, h: int, x1: int, y1: int, x2: int, y2: int):
    """cv::clipLine on the image rectangle: returns the clipped end points or None when the segment misses it."""
    right, bottom = w - 1, h - 1
    if w <= 0 or h <= 0:
        return None

    def code_full(x, y):
        return (x < 0) + (x > right) * 2 + (y < 0) * 4 + (y > bottom) * 8

    c1, c2 = code_full(x1, y1), code_full(x2, y2)
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (x1, y1, x2, y2) if (c1 | c2) == 0 else None


def _draw_line8(mask: np.ndarray, p1: Point, p2: Point) -> None:
    """cv::Line with connectivity 8: clip, then the LineIterator's Bresenham walk from the LEFT end point
    (leftToRight = true): one pixel per step of the major axis, the minor axis advancing when err < 0."""
    h, w = mask.shape
    c = _clip_line(w, h, int(p1[0]), int(p1[1]), int(p2[0]), int(p2[1]))
    if c is None:
        return
    x1, y1, x2, y2 = c
    if x2 < x1:
        x1, y1, x2, y2 = x2, y2, x1, y1
    dx, dy = x2 - x1, y2 - y1
    sy = -1 if dy < 0 else 1
    dy = abs(dy)
    if dy > dx:                 # y is the major axis
        major, minor, step_major, step_minor = dy, dx, (0, sy), (1, 0)
    else:
        major, minor, step_major, step_minor = dx, dy, (1, 0), (0, sy)
    err = major - 2 * minor
    x, y = x1, y1
    for _ in range(major + 1):
        mask[y, x] = 255
        neg = err < 0
        err += -2 * minor + (2 * major if neg else 0)
        x += step_major[0] + (step_minor[0] if neg else 0)
        y += step_major[1] + (step_minor[1] if neg else 0)


def fill_poly_mask(height: int, width: int, points: Sequence[Point]) -> np.ndarray:
    """``mask = zeros((height, width), uint8); cv2.fillPoly(mask, [points], 255)`` for integer points.

    A pixel is set when it lies on one of the polygon's edges as the 8-connected line draws it, or inside the polygon
    by the even-odd scan-line rule: on every row y in [y_min, y_max) the active edges, sorted by their x at that row
    (16.16 fixed point, slope = truncated (x1 - x0) * 65536 / (y1 - y0), advanced by one slope per row), are paired
    and the span [ceil(x_left), floor(x_right)] is filled.  Horizontal edges only draw their line.  Points may lie
    outside the image (lines and spans are clipped)."""
    mask = np.zeros((height, width), dtype=np.uint8)
    pts = [(int(p[0]), int(p[1])) for p in points]
    n = len(pts)
    if n == 0 or height <= 0 or width <= 0:
        return mask
    edges: List[List[int]] = []            # [y0, y1, x (fixed point at y0), dx]
    p0 = pts[-1]
    for p1 in pts:
        _draw_line8(mask, p0, p1)
        if p0[1] != p1[1]:
            dxf = _trunc_div((p1[0] - p0[0]) << XY_SHIFT, p1[1] - p0[1])
            if p0[1] < p1[1]:
                edges.append([p0[1], p1[1], p0[0] << XY_SHIFT, dxf])
            else:
                edges.append([p1[1], p0[1], p1[0] << XY_SHIFT, dxf])
        p0 = p1
    if not edges:
        return mask
    y_min = min(e[0] for e in edges)
    y_max = min(max(e[1] for e in edges), height)
    for y in range(y_min, y_max):
        active = []
        for e in edges:
            if e[0] <= y < e[1]:
                active.append(e[2] + (y - e[0]) * e[3])
        if y < 0:
            continue
        active.sort()
        for i in range(0, len(active) - 1, 2):
            xl, xr = active[i], active[i + 1]
            x1 = (xl + XY_ONE - 1) >> XY_SHIFT
            x2 = xr >> XY_SHIFT
            if x1 < width and x2 >= 0:
                x1 = max(x1, 0)
                x2 = min(x2, width - 1)
                if x2 >= x1:
                    mask[y, x1:x2 + 1] = 255
    return mask


def polygon_white_fill(crop: np.ndarray, rel_points: Sequence[Point]) -> np.ndarray:
    """``src/ui/main_window.py:6499-6506``: pixels of the crop inside the polygon, white everywhere else
    (``fg = crop & mask``, ``bg = 255 & ~mask``, ``fg + bg`` saturating - the two are disjoint, so a select)."""
    mask = fill_poly_mask(crop.shape[0], crop.shape[1], rel_points)
    out = np.full_like(crop, 255)
    sel = mask != 0
    out[sel] = crop[sel]
    return out


def rect_crop_bgr(page_rgb: np.ndarray, rect: Tuple[int, int, int, int]) -> Optional[np.ndarray]:
    """``process_rect_area`` (``:6428-6431``): PIL ``crop((x, y, rect.right(), rect.bottom()))`` of the RGB page,
    i.e. columns [x, x + w - 1) and rows [y, y + h - 1), zero-filled outside the page like PIL, then RGB -> BGR."""
    x, y, w, h = (int(v) for v in rect)
    if w <= 0 or h <= 0:
        return None
    cw, ch = w - 1, h - 1
    if cw <= 0 or ch <= 0:
        return None
    H, W = page_rgb.shape[:2]
    out = np.zeros((ch, cw, 3), dtype=np.uint8)
    sx0, sy0, sx1, sy1 = max(x, 0), max(y, 0), min(x + cw, W), min(y + ch, H)
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y:sy1 - y, sx0 - x:sx1 - x] = page_rgb[sy0:sy1, sx0:sx1, :3]
    return np.ascontiguousarray(out[..., ::-1])


def polygon_crop_bgr(page_rgb: np.ndarray, polygon: Sequence[Point], bbox: Optional[Tuple[int, int, int, int]] = None) -> Optional[np.ndarray]:
    """``process_confirmed_polygon`` (``:6481-6506``): the bounding-box crop (as ``rect_crop_bgr``) with everything
    outside the polygon painted white; polygon points are taken relative to the box origin."""
    if bbox is None:
        bbox = bounding_rect(polygon)
    crop = rect_crop_bgr(page_rgb, bbox)
    if crop is None:
        return None
    rel = [(int(px) - bbox[0], int(py) - bbox[1]) for px, py in polygon]
    return polygon_white_fill(crop, rel)


# ------------------------------------------------------------------------------------------ Text-detect driver
Detection = Dict[str, object]


def region_rects(polygons: Iterable[Sequence[Point]]) -> List[Tuple[int, int, int, int]]:
    return [bounding_rect(p) for p in polygons]


def recognize_pages(reader, pages_bgr: Sequence[np.ndarray], regions_per_page: Sequence[Sequence[Tuple[str, Sequence[Point]]]],
                    on_error=None) -> List[List[Detection]]:
    """All regions of all pages as ONE engine job queue (the reference does pages x regions serial B = 1 calls:
    ``AutoDetectorWorker.run`` -> ``_collect_manga_detections`` -> ``_recognize_polygon``).

    ``reader``: a ``MangaOcr``; ``pages_bgr[i]``: uint8 [H,W,3] page as ``cv2.cvtColor(np.array(pil), COLOR_RGB2BGR)``
    gives it (``workers.py:460-461``); ``regions_per_page[i]``: the detector's ``(text, polygon)`` pairs for that page.
    Returns per page ``[{'polygon': polygon, 'text': recognized.strip() or text}]`` in region order - exactly the list
    ``detection_complete`` carries for Manga-OCR in Text mode, before the application's own noise filter and merging.
    A page that fails validation is reported through ``on_error(page_index, exc)`` and yields ``[]``
    (``main_window.py:9299-9303``, ``workers.py:477-479``); the other pages are unaffected."""
    if len(pages_bgr) != len(regions_per_page):
        raise ValueError("one region list per page")
    good_pages, page_slot, flat = [], {}, []
    out: List[List[Detection]] = [[] for _ in pages_bgr]
    for pi, (page, regs) in enumerate(zip(pages_bgr, regions_per_page)):
        try:
            a = np.asarray(page)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3 or a.shape[0] < 1 or a.shape[1] < 1:
                raise ValueError("page must be a uint8 [H,W,3] BGR array")
            regs = list(regs)
            rects = region_rects(poly for _, poly in regs)
        except Exception as exc:      # noqa: BLE001 - per-page isolation, like the reference's loop
            if on_error is not None:
                on_error(pi, exc)
            continue
        page_slot[pi] = len(good_pages)
        good_pages.append(a)
        for ri, ((text, poly), rect) in enumerate(zip(regs, rects)):
            flat.append((pi, ri, text, poly, rect))
        out[pi] = [None] * len(regs)
    texts = reader.recognize_regions(good_pages, [(page_slot[pi],) + rect for pi, _, _, _, rect in flat]) if flat else []
    for (pi, ri, text, poly, _), rec in zip(flat, texts):
        out[pi][ri] = {"polygon": poly, "text": (rec.strip() or text)}
    return out


def recognize_page(reader, page_bgr: np.ndarray, regions: Sequence[Tuple[str, Sequence[Point]]]) -> List[Detection]:
    """One page: ``_collect_manga_detections`` with every region in one batch."""
    return recognize_pages(reader, [page_bgr], [regions])[0]
