"""CPU: the strip schedule of the persistent encoder GEMM (csrc/kernels_gemm_pers.h, STRIP; host mirror pers_strip_rows /
pers_strip_wins in csrc/engine.hip) restated in Python and checked as a PROPERTY over many shapes: whatever M, the number
of 256-column slices and the grid, every (row, slice) of the output is owned by exactly one tile of exactly one block, no
tile's window reads a row behind M, a half tile owns at most 128 rows and a block has at most one.  The GPU tests
(tests/test_gpu_kernels.py::test_gemm_persistent_strip_schedule*) check the kernel itself at a handful of shapes; this
pins the arithmetic they share."""
import numpy as np
import pytest


def strip_blocks(M, ntn, grid):
    """-> list of (block, col, [(m0, lo, hi, half), ...]) exactly as the kernel derives them from blockIdx.x."""
    out = []
    slots = grid >> 3
    spx = slots // ntn
    used = spx * ntn
    nstrips = 8 * spx + (8 * (slots - used)) // ntn
    for b in range(grid):
        xcd, k = b & 7, b >> 3
        if k < used:
            strip, col = xcd * spx + k // ntn, k % ntn
        else:
            j = (k - used) * 8 + xcd
            strip, col = 8 * spx + j // ntn, j % ntn
        if strip >= nstrips:
            continue
        U = (M + 15) >> 4
        base, extra = U // nstrips, U % nstrips
        u0 = strip * base + min(strip, extra)
        nu = base + (1 if strip < extra else 0)
        rs, re = u0 * 16, min(M, (u0 + nu) * 16)
        if re <= rs:
            continue
        R = re - rs
        rem = R & 255
        ntl = (R >> 8) + (1 if rem else 0)
        hpos = -1
        if rem and rem <= 128:
            v = strip % 3
            hpos = ntl - 1 if v == 0 else 0 if v == 1 else ntl >> 1
        tiles = []
        for t in range(ntl):
            half = t == hpos
            lo = rs + 256 * t - ((256 - rem) if (hpos >= 0 and t > hpos) else 0)
            hi = min(re, lo + (rem if half else 256))
            m0 = max(0, min(lo, hi - (128 if half else 256)))
            tiles.append((m0, lo, hi, half))
        out.append((b, col, tiles))
    return out


@pytest.mark.parametrize("ntn", [1, 2, 3, 4, 9, 12])
@pytest.mark.parametrize("grid", [8, 16, 64, 256, 304])
def test_every_row_of_every_slice_is_owned_exactly_once(ntn, grid):
    rs = np.random.RandomState(ntn * 1000 + grid)
    Ms = [50432, 33000, 8900, 2500, 2200, 1000, 300, 256, 257, 4096] + [int(v) for v in rs.randint(129, 70000, size=25)]
    for M in Ms:
        blocks = strip_blocks(M, ntn, grid)
        slots = grid >> 3
        if 8 * (slots // ntn) + (8 * (slots - (slots // ntn) * ntn)) // ntn == 0:
            assert not blocks          # the grid cannot hold a strip: the host never picks the schedule (pers_strip_rows == 0)
            continue
        owned = np.zeros((ntn, M), dtype=np.int32)
        seen = set()
        for b, col, tiles in blocks:
            assert b not in seen and 0 <= col < ntn
            seen.add(b)
            assert sum(1 for t in tiles if t[3]) <= 1, "more than one half tile in a strip"
            for m0, lo, hi, half in tiles:
                win = 128 if half else 256
                assert 0 <= m0 <= lo < hi <= M and hi <= m0 + win, (M, ntn, grid, m0, lo, hi, half)
                assert m0 + win <= max(M, win), "the window reads a row behind M"
                if half:
                    assert hi - lo <= 128
                owned[col, lo:hi] += 1
        assert (owned == 1).all(), f"M={M} ntn={ntn} grid={grid}: rows owned {owned.min()}..{owned.max()} times"


def test_the_encoder_shape_walks_two_tiles_and_a_half_tile_per_block():
    """Batch 256: 50,432 rows, N = 768 -> 85 strips of 592 / 608 rows on 255 of 256 blocks; the half tile first, last or in
    the middle by strip number (the blocks' epilogues are staggered)."""
    blocks = strip_blocks(50432, 3, 256)
    assert len(blocks) == 255
    shapes = {tuple((hi - lo, half) for _, lo, hi, half in tiles) for _, _, tiles in blocks}
    assert shapes <= {((256, False), (256, False), (80, True)), ((256, False), (256, False), (96, True)),
                      ((80, True), (256, False), (256, False)), ((96, True), (256, False), (256, False)),
                      ((256, False), (80, True), (256, False)), ((256, False), (96, True), (256, False))}
    assert len({s.index(next(x for x in s if x[1])) for s in shapes}) == 3
    # the three column blocks of a strip are neighbours in one XCD's block order (b, b + 8, b + 16), 10 strips per XCD
    by_rows = {}
    for b, col, tiles in blocks:
        by_rows.setdefault(tiles[0][1] if not tiles[0][3] else tiles[0][1], []).append((b, col))
    xcd_local = sum(1 for v in by_rows.values() if len({b & 7 for b, _ in v}) == 1)
    assert xcd_local >= 80
