"""CPU: the crop-job producers and the Text-detect page driver.

``cv2`` is not in the image, so ``fill_poly_mask`` is pinned by fixtures derived by hand from OpenCV's rule for
``fillPoly(mask, [pts], 255)`` with integer points: the polygon's edges as 8-connected Bresenham lines (drawn from the
left end point) plus the even-odd interior spans [ceil(x_left), floor(x_right)] of every row in [y_min, y_max)."""
import numpy as np
import pytest

from manga_ocr import regions as R


def grid(rows):
    return np.array([[255 if c == "#" else 0 for c in r] for r in rows], dtype=np.uint8)


def test_fill_poly_axis_aligned_rectangle_includes_its_border():
    m = R.fill_poly_mask(6, 8, [(1, 1), (5, 1), (5, 4), (1, 4)])
    want = np.zeros((6, 8), np.uint8)
    want[1:5, 1:6] = 255
    np.testing.assert_array_equal(m, want)


def test_fill_poly_right_triangle_and_diamond():
    m = R.fill_poly_mask(7, 7, [(0, 0), (6, 0), (0, 6)])
    yy, xx = np.mgrid[0:7, 0:7]
    np.testing.assert_array_equal(m, np.where(xx + yy <= 6, 255, 0).astype(np.uint8))
    d = R.fill_poly_mask(7, 7, [(3, 0), (6, 3), (3, 6), (0, 3)])
    np.testing.assert_array_equal(d, np.where(abs(xx - 3) + abs(yy - 3) <= 3, 255, 0).astype(np.uint8))


def test_fill_poly_shallow_edge_follows_the_bresenham_line():
    # hypotenuse (0,0)-(8,3): line pixels (0,0)(1,0)(2,1)(3,1)(4,1)(5,2)(6,2)(7,3)(8,3); interior spans start at
    # ceil(8y/3): row 1 -> 3, row 2 -> 6; the union per row:
    m = R.fill_poly_mask(4, 9, [(0, 0), (8, 0), (8, 3)])
    np.testing.assert_array_equal(m, grid(["#########",
                                            "..#######",
                                            ".....####",
                                            ".......##"]))


def test_fill_poly_concave_l_shape_uses_the_even_odd_rule():
    m = R.fill_poly_mask(6, 6, [(0, 0), (5, 0), (5, 2), (2, 2), (2, 5), (0, 5)])
    want = np.zeros((6, 6), np.uint8)
    want[0:3, 0:6] = 255
    want[0:6, 0:3] = 255
    np.testing.assert_array_equal(m, want)


def test_fill_poly_clips_to_the_image_and_handles_degenerate_input():
    m = R.fill_poly_mask(6, 6, [(2, 2), (9, 2), (9, 9), (2, 9)])
    want = np.zeros((6, 6), np.uint8)
    want[2:, 2:] = 255
    np.testing.assert_array_equal(m, want)
    assert R.fill_poly_mask(4, 4, []).sum() == 0
    np.testing.assert_array_equal(R.fill_poly_mask(3, 3, [(1, 1)]), grid(["...", ".#.", "..."]))          # a point
    np.testing.assert_array_equal(R.fill_poly_mask(3, 4, [(0, 1), (3, 1)]), grid(["....", "####", "...."]))  # a segment
    assert R.fill_poly_mask(4, 4, [(-5, -5), (-2, -5), (-2, -2)]).sum() == 0                               # all outside


def test_bounding_rect_is_qt_inclusive():
    assert R.bounding_rect([(3, 4), (10, 4), (10, 9), (3, 9)]) == (3, 4, 8, 6)
    assert R.bounding_rect([(5, 5)]) == (5, 5, 1, 1)
    assert R.bounding_rect([]) == (0, 0, 0, 0)


def test_rect_and_polygon_crop_jobs_follow_the_reference_boxes():
    rs = np.random.RandomState(5)
    page = rs.randint(0, 256, size=(40, 50, 3), dtype=np.uint8)       # RGB page (current_image_pil)
    # process_rect_area: crop((x, y, right(), bottom())) = (w-1) x (h-1) pixels, then RGB -> BGR
    c = R.rect_crop_bgr(page, (10, 5, 21, 11))
    assert c.shape == (10, 20, 3)
    np.testing.assert_array_equal(c, page[5:15, 10:30, ::-1])
    assert R.rect_crop_bgr(page, (10, 5, 1, 9)) is None and R.rect_crop_bgr(page, (0, 0, 0, 0)) is None
    # PIL zero-fills what lies outside the page
    e = R.rect_crop_bgr(page, (45, 35, 11, 11))
    assert e.shape == (10, 10, 3) and (e[5:, :] == 0).all() and (e[:, 5:] == 0).all()
    np.testing.assert_array_equal(e[:5, :5], page[35:40, 45:50, ::-1])
    # process_confirmed_polygon: a triangle; white outside, page pixels inside; the box is the polygon's boundingRect
    tri = [(10, 5), (30, 5), (10, 25)]
    bbox = R.bounding_rect(tri)
    assert bbox == (10, 5, 21, 21)
    p = R.polygon_crop_bgr(page, tri)
    assert p.shape == (20, 20, 3)
    yy, xx = np.mgrid[0:20, 0:20]
    inside = xx + yy <= 20
    np.testing.assert_array_equal(p[inside], page[5:25, 10:30, ::-1][inside])
    assert (p[~inside] == 255).all()
    # every pixel of the rectangle polygon of its own box is kept
    box_poly = [(10, 5), (30, 5), (30, 25), (10, 25)]
    np.testing.assert_array_equal(R.polygon_crop_bgr(page, box_poly), page[5:25, 10:30, ::-1])


class FakeReader:
    """Stands in for MangaOcr.recognize_regions: answers with the region's rectangle, so the test sees what the
    driver sent; one region answers '' to exercise `recognized or text`."""

    def __init__(self):
        self.calls = []

    def recognize_regions(self, pages, regions):
        self.calls.append((len(pages), list(regions)))
        return ["" if r[3] == 7 else f" p{r[0]}:{r[1]},{r[2]},{r[3]}x{r[4]} " for r in regions]


def test_recognize_pages_batches_every_region_of_every_page_in_one_call():
    pages = [np.zeros((100, 80, 3), np.uint8), np.zeros((60, 60, 3), np.uint8), np.zeros((10, 10), np.uint8)]
    sq = lambda x, y, w, h: [(x, y), (x + w - 1, y), (x + w - 1, y + h - 1), (x, y + h - 1)]    # noqa: E731
    regs = [[("a", sq(5, 5, 20, 30)), ("kept", sq(1, 1, 7, 7))], [], [("c", sq(0, 0, 4, 4))]]
    errs = []
    rd = FakeReader()
    out = R.recognize_pages(rd, pages, regs, on_error=lambda i, e: errs.append((i, str(e))))
    assert len(rd.calls) == 1 and rd.calls[0][0] == 2          # the malformed third page never reaches the engine
    assert rd.calls[0][1] == [(0, 5, 5, 20, 30), (0, 1, 1, 7, 7)]
    assert out[0] == [{"polygon": regs[0][0][1], "text": "p0:5,5,20x30"}, {"polygon": regs[0][1][1], "text": "kept"}]
    assert out[1] == [] and out[2] == [] and errs and errs[0][0] == 2
    assert R.recognize_page(rd, pages[1], [("z", sq(2, 2, 9, 9))]) == [{"polygon": sq(2, 2, 9, 9), "text": "p0:2,2,9x9"}]
    with pytest.raises(ValueError):
        R.recognize_pages(rd, pages, regs[:2])
