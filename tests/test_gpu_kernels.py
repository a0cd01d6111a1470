"""-m gpu: every HIP kernel class against a float64 numpy / torch-fp32 statement of the same op,
called through the C ABI's operator hooks.  Device memory comes from torch (plumbing only)."""
import os

import numpy as np
import pytest

from gpu_util import bf16_round, engine, report

# tile codes 256 .. 2048 and 4098 are the A/B kernels of rounds 1-2: they exist in the experiments build only
# (python manga-ocr_amd/build.py --experiments; MOCR_LIB=.../libmocr_hip_lab.so) - the product library ships one kernel per role
LAB = "lab" in os.path.basename(os.environ.get("MOCR_LIB", ""))

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

EPI_SLAB, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_PATCH, EPI_BIAS_F32 = range(6)


def _dev(a, dtype):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(torch.bfloat16) if dtype == "bf16" else t


def _gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("tile,M,N,K", [(128, 256, 256, 128), (128, 384, 768, 768), (64, 64, 768, 768),
                                         (64, 192, 128, 3072), (128, 394, 2304, 768)])
@pytest.mark.parametrize("epi", [EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_BIAS_F32])
def test_gemm_epilogues(dtype, tile, M, N, K, epi):
    eng = engine(dtype)
    rs = np.random.RandomState(M + N + K + epi)
    Mp = (M + tile - 1) // tile * tile                       # the kernel stages whole tiles of A
    A = rs.standard_normal((Mp, K)).astype(np.float32)
    W = (rs.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rs.standard_normal(N).astype(np.float32)
    resid = rs.standard_normal((M, N)).astype(np.float32)
    if dtype == "bf16":
        A, W = bf16_round(A), bf16_round(W)
    ref = A[:M].astype(np.float64) @ W.astype(np.float64).T + bias
    if epi == EPI_BIAS_GELU:
        ref = _gelu(ref)
    if epi == EPI_BIAS_RESID:
        ref = ref + resid
    dA, dW = _dev(A, dtype), _dev(W, dtype)
    dB = torch.from_numpy(bias).cuda()
    out_f32 = epi in (EPI_BIAS_RESID, EPI_BIAS_F32)
    dO = torch.full((M, N), float("nan"), device="cuda",
                    dtype=torch.float32 if (out_f32 or dtype == "fp32") else torch.bfloat16)
    dR = torch.from_numpy(resid).cuda() if epi == EPI_BIAS_RESID else None
    torch.cuda.synchronize()
    eng.op_gemm(dA, dW, dB, dO, dR, M, N, K, epi, tile=tile, split_k=1)
    got = dO.float().cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max() / scale
    tol = 2e-6 * np.sqrt(K) if (dtype == "fp32") else (1e-5 if out_f32 else 6e-3)
    report(f"gemm {dtype} tile{tile} M{M} N{N} K{K} epi{epi}: max rel err {err:.3e} (tol {tol:.1e})")
    assert np.isfinite(got).all()
    assert err <= tol


def _big_tile_cases():
    """(M, N, K, epi, tile) the big-tile kernels support: the A/B kernels of rounds 1-2 (tile codes 256 .. 2048) only in the
    experiments build (MOCR_LIB=.../libmocr_hip_lab.so) - in the product run they are not collected at all."""
    out = []
    for tile in ([256, 512, 1024, 2048] if LAB else []) + [4096, 4097]:
        for epi in (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_BIAS_F32):
            for M, N, K in [(256, 256, 64), (512, 768, 768), (700, 2304, 768), (1000, 768, 3072), (256, 768, 256), (300, 512, 128)]:
                if tile >= 512 and epi == EPI_BIAS_F32:      # the wide kernels have the encoder layers' epilogues only
                    continue
                if tile >= 2048 and K < 128:                  # the four-stage kernels (fragments requested across the K-tile barrier) need >= 4 K-tiles
                    continue
                if tile < 2048 and K == 128:                  # shape added for the four-stage kernels' shortest K loop
                    continue
                out.append((M, N, K, epi, tile))
    return out


@pytest.mark.parametrize("M,N,K,epi,tile", _big_tile_cases())
def test_gemm_256x128_three_stage_ring(M, N, K, epi, tile):
    """The big-tile encoder kernels (bf16 only; 256 = 64-deep K-tiles, epilogue through LDS; 512 = the
    "wide" kernel: 32-deep K-tiles, two blocks per CU, epilogue from the registers): K from 1 to 96
    K-tiles exercises the ring's prologue, steady state (counted vmcnt) and drain; M not a multiple of
    256 exercises the row guard."""
    eng = engine("bf16")
    rs = np.random.RandomState(M + N + K + epi)
    Mp = (M + 255) // 256 * 256
    A = bf16_round(rs.standard_normal((Mp, K)).astype(np.float32))
    W = bf16_round((rs.standard_normal((N, K)) * 0.05).astype(np.float32))
    bias = rs.standard_normal(N).astype(np.float32)
    resid = rs.standard_normal((M, N)).astype(np.float32)
    ref = A[:M].astype(np.float64) @ W.astype(np.float64).T + bias
    if epi == EPI_BIAS_GELU:
        ref = _gelu(ref)
    if epi == EPI_BIAS_RESID:
        ref = ref + resid
    out_f32 = epi in (EPI_BIAS_RESID, EPI_BIAS_F32)
    dO = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)
    dR = torch.from_numpy(resid).cuda() if epi == EPI_BIAS_RESID else None
    dA, dW, dB = _dev(A, "bf16"), _dev(W, "bf16"), torch.from_numpy(bias).cuda()
    torch.cuda.synchronize()
    eng.op_gemm(dA, dW, dB, dO, dR, M, N, K, epi, tile=tile, split_k=1)
    got = dO.float().cpu().numpy().astype(np.float64)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    tol = 1e-5 if out_f32 else 6e-3
    report(f"gemm tile{tile} M{M} N{N} K{K} epi{epi}: max rel err {err:.3e} (tol {tol:.1e})")
    assert np.isfinite(got).all() and err <= tol


@pytest.mark.parametrize("epi", [EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID])
@pytest.mark.parametrize("tile,M,N,K", [(4096, 8900, 2304, 768), (4097, 2500, 768, 3072), (4097, 1300, 3072, 128), (4096, 33000, 768, 768)] +
                         ([(4098, 8900, 2304, 768), (4101, 8900, 2304, 768), (4102, 2500, 768, 3072), (4102, 1300, 3072, 128),
                           (4105, 8900, 2304, 768), (4106, 2500, 768, 3072), (4106, 1300, 3072, 128)] if LAB else []))
def test_gemm_persistent_blocks_walk_several_tiles(epi, tile, M, N, K):
    """gemm_pers_kernel (tile code 4096; 4097 = the same kernel on 8 blocks): a block multiplies a SEQUENCE of 256 x 256
    tiles - the LDS ring runs across tile boundaries, the epilogue is staged through the slot the last K-tile left, and
    only four of the eight waves touch global memory - so shapes with more tiles than blocks are what tests it: 315 tiles
    on 256 blocks, 30 / 72 tiles on 8 blocks (3-9 tiles per block, K loops of 4 and 96 K-tiles), 387 tiles of N = 768.
    Every output element is checked against float64."""
    eng = engine("bf16")
    rs = np.random.RandomState(M + N + K + epi)
    Mp = (M + 255) // 256 * 256
    A = bf16_round(rs.standard_normal((Mp, K)).astype(np.float32))
    W = bf16_round((rs.standard_normal((N, K)) * 0.05).astype(np.float32))
    bias = rs.standard_normal(N).astype(np.float32)
    resid = rs.standard_normal((M, N)).astype(np.float32) if epi == EPI_BIAS_RESID else None
    ref = A[:M].astype(np.float64) @ W.astype(np.float64).T + bias
    if epi == EPI_BIAS_GELU:
        ref = _gelu(ref)
    if epi == EPI_BIAS_RESID:
        ref = ref + resid
    out_f32 = epi == EPI_BIAS_RESID
    dO = torch.full((M + 3, N), float("nan"), device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)   # 3 guard rows behind the output
    dR = torch.from_numpy(resid).cuda() if resid is not None else None
    dA, dW, dB = _dev(A, "bf16"), _dev(W, "bf16"), torch.from_numpy(bias).cuda()
    torch.cuda.synchronize()
    for rep in range(2):                      # twice: the second launch finds the caches warm and the blocks in another phase
        eng.op_gemm(dA, dW, dB, dO, dR, M, N, K, epi, tile=tile, split_k=1)
        got = dO.float().cpu().numpy().astype(np.float64)
        assert np.isnan(got[M:]).all(), "rows behind M were written"
        err = np.abs(got[:M] - ref).max() / np.abs(ref).max()
        tol = 1e-5 if out_f32 else 6e-3
        assert np.isfinite(got[:M]).all() and err <= tol, f"rep {rep}: max rel err {err:.3e}"
    report(f"persistent gemm tile{tile} M{M} N{N} K{K} epi{epi}: max rel err {err:.3e} (tol {tol:.1e})")


@pytest.mark.parametrize("tile,M,N,K", [(4099, 50432, 768, 768), (4099, 33000, 768, 768), (4099, 8900, 2304, 768), (4099, 20000, 768, 3072),
                                        (4100, 2500, 768, 3072), (4100, 2200, 768, 128), (4100, 1000, 1024, 256), (4100, 300, 768, 256)] +
                         ([(4103, 50432, 768, 3072), (4103, 33000, 768, 768), (4104, 2500, 768, 3072), (4104, 2200, 768, 128)] if LAB else []))
def test_gemm_persistent_strip_schedule(tile, M, N, K):
    """gemm_pers_kernel's strip schedule (tile code 4099; 4100 = on 8 blocks), the fp32-residual GEMMs of the encoder: a block
    owns a strip of rows of one 256-column slice - 256-row tiles plus one HALF tile (a wave multiplies 64 x 64) first, in the
    middle or last, windows shifted up where a tile is short.  50,432 x 768: the encoder's own shape (85 strips of 592 / 608
    rows: two tiles + a half tile of 80 / 96 rows); 33,000 rows: remainders of 128 (half) and 144 rows (short full tile), M
    not a multiple of 16; N = 2304: strips made of the XCDs' left-over blocks; 8 blocks: two or three strips of 4-5 tiles;
    300 rows: one short tile per strip, its window shifted up.  In place (out = resid, as the encoder calls it); every element against float64."""
    eng = engine("bf16")
    rs = np.random.RandomState(M + N + K + tile)
    A = bf16_round(rs.standard_normal((M, K)).astype(np.float32))          # NOT padded: the strip schedule reads no row behind M
    W = bf16_round((rs.standard_normal((N, K)) * 0.05).astype(np.float32))
    bias = rs.standard_normal(N).astype(np.float32)
    resid = rs.standard_normal((M, N)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + bias + resid
    dA, dW, dB = _dev(A, "bf16"), _dev(W, "bf16"), torch.from_numpy(bias).cuda()
    for rep in range(2):
        dO = torch.full((M + 3, N), float("nan"), device="cuda", dtype=torch.float32)      # 3 guard rows behind the output
        dO[:M] = torch.from_numpy(resid).cuda()
        torch.cuda.synchronize()
        eng.op_gemm(dA, dW, dB, dO, dO, M, N, K, EPI_BIAS_RESID, tile=tile, split_k=1)
        got = dO.float().cpu().numpy().astype(np.float64)
        assert np.isnan(got[M:]).all(), "rows behind M were written"
        err = np.abs(got[:M] - ref).max() / np.abs(ref).max()
        assert np.isfinite(got[:M]).all() and err <= 1e-5, f"rep {rep}: max rel err {err:.3e}"
    report(f"persistent gemm, strips, tile{tile} M{M} N{N} K{K}: max rel err {err:.3e} (tol 1.0e-05)")


@pytest.mark.parametrize("epi", [EPI_BIAS, EPI_BIAS_GELU])
@pytest.mark.parametrize("tile,M,N,K", [(4099, 50432, 3072, 768), (4099, 33000, 2304, 768), (4100, 2200, 768, 128), (4100, 1000, 1024, 256), (4100, 300, 768, 256)])
def test_gemm_persistent_strip_schedule_bf16_outputs(epi, tile, M, N, K):
    """The strip schedule with the bf16 epilogues (FC1 at batch 256: 9.23 rounds of tiles walked as 9 tiles + a half tile):
    a half tile is staged and stored in ONE pass of 128 rows."""
    eng = engine("bf16")
    rs = np.random.RandomState(M + N + K + epi + tile)
    A = bf16_round(rs.standard_normal((M, K)).astype(np.float32))          # not padded: strips read no row behind M
    W = bf16_round((rs.standard_normal((N, K)) * 0.05).astype(np.float32))
    bias = rs.standard_normal(N).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + bias
    if epi == EPI_BIAS_GELU:
        ref = _gelu(ref)
    dA, dW, dB = _dev(A, "bf16"), _dev(W, "bf16"), torch.from_numpy(bias).cuda()
    dO = torch.full((M + 3, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize()
    for rep in range(2):
        eng.op_gemm(dA, dW, dB, dO, None, M, N, K, epi, tile=tile, split_k=1)
        got = dO.float().cpu().numpy().astype(np.float64)
        assert np.isnan(got[M:]).all(), "rows behind M were written"
        err = np.abs(got[:M] - ref).max() / np.abs(ref).max()
        assert np.isfinite(got[:M]).all() and err <= 6e-3, f"rep {rep}: max rel err {err:.3e}"
    report(f"persistent gemm, strips, tile{tile} M{M} N{N} K{K} epi{epi}: max rel err {err:.3e} (tol 6.0e-03)")


def _ln_stats_ref(x64):
    """[M, D] float64 -> partial layout of one 256-column slice per slot: [M, 4, 2] (sum, sum of squares)."""
    M, D = x64.shape
    part = np.zeros((M, 4, 2))
    for t in range(D // 256):
        seg = x64[:, 256 * t:256 * (t + 1)]
        part[:, t, 0] = seg.sum(1)
        part[:, t, 1] = (seg * seg).sum(1)
    return part


@pytest.mark.parametrize("tile,M,K", [(4096, 33000, 768), (4099, 50432, 768), (4099, 33000, 3072), (4100, 2200, 768), (4097, 2500, 3072)])
def test_gemm_persistent_residual_emits_ln_operands(tile, M, K):
    """LayerNorm folded into the encoder GEMMs, producer side (gemm_pers_kernel LNF, EPI_BIAS_RESID; tile list and strips):
    next to out = resid + A W^T + b the kernel writes the rows as bf16 and each row's (sum, sum of squares) per 256-column
    slice - what the GEMM behind the LayerNorm needs instead of LN(out)."""
    eng = engine("bf16")
    N = 768
    rs = np.random.RandomState(M + K + tile)
    Mp = (M + 255) // 256 * 256
    A = bf16_round(rs.standard_normal((Mp, K)).astype(np.float32))
    W = bf16_round((rs.standard_normal((N, K)) * 0.05).astype(np.float32))
    bias = rs.standard_normal(N).astype(np.float32)
    resid = (rs.standard_normal((M, N)) * 2.0 + rs.standard_normal((M, 1))).astype(np.float32)      # rows with a mean of their own
    ref = A[:M].astype(np.float64) @ W.astype(np.float64).T + bias + resid
    dA, dW, dB = _dev(A, "bf16"), _dev(W, "bf16"), torch.from_numpy(bias).cuda()
    dO = torch.full((M + 3, N), float("nan"), device="cuda", dtype=torch.float32)
    dO[:M] = torch.from_numpy(resid).cuda()
    dXb = torch.full((M + 3, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    dP = torch.full((M + 3, 4, 2), -7.0, device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    eng.op_gemm_ln(dA, dW, dB, dO, dO, M, N, K, EPI_BIAS_RESID, tile, dP, None, dXb)
    got = dO.cpu().numpy().astype(np.float64)
    assert np.isnan(got[M:]).all(), "rows behind M were written"
    err = np.abs(got[:M] - ref).max() / np.abs(ref).max()
    assert err <= 1e-5, f"out: max rel err {err:.3e}"
    xb = dXb.float().cpu().numpy()
    assert np.isnan(xb[M:]).all()
    assert np.array_equal(xb[:M], bf16_round(got[:M].astype(np.float32))), "the bf16 copy is not the rounded fp32 output"
    part = dP.cpu().numpy().astype(np.float64)
    assert (part[M:] == -7.0).all() and (part[:M, 3] == -7.0).all(), "statistics written outside rows < M / slices 0..2"
    want = _ln_stats_ref(got[:M])
    perr = np.abs(part[:M, :3] - want[:, :3]).max(axis=(0, 1)) / np.abs(want[:, :3]).max(axis=(0, 1))
    report(f"persistent gemm + LN operands tile{tile} M{M} K{K}: out {err:.2e}, sums {perr[0]:.2e}, squares {perr[1]:.2e}")
    assert perr.max() <= 2e-6


@pytest.mark.parametrize("epi", [EPI_BIAS, EPI_BIAS_GELU])
@pytest.mark.parametrize("tile,M,N", [(4096, 8900, 2304), (4097, 2500, 3072), (4096, 50432, 2304), (4099, 50432, 3072), (4100, 2200, 768)])
def test_gemm_persistent_with_folded_layernorm(epi, tile, M, N):
    """Consumer side (EPI_BIAS / EPI_BIAS_GELU with LNF): A = bf16 x, W = bf16(W o gamma), bias = b + W beta, column sums and the
    rows' statistics -> LN(x) W^T + b.  Checked (a) against the identity evaluated in float64 on the operands the kernel
    was given (bf16 output rounding only) and (b) against the real thing, LN(x) in float64 times the unrounded weight."""
    eng = engine("bf16")
    K = 768
    rs = np.random.RandomState(M + N + epi)
    x = (rs.standard_normal((M, K)) * (0.5 + rs.rand(M, 1) * 3) + rs.standard_normal((M, 1)) * 0.7).astype(np.float32)
    x[:, 5] *= 8.0                                                     # an outlier channel
    g = (1.0 + 0.2 * rs.standard_normal(K)).astype(np.float32)
    be = (0.1 * rs.standard_normal(K)).astype(np.float32)
    W = (rs.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = rs.standard_normal(N).astype(np.float32)
    Wf = bf16_round(W * g)
    csum = Wf.astype(np.float64).sum(1).astype(np.float32)
    bf = (b.astype(np.float64) + W.astype(np.float64) @ be.astype(np.float64)).astype(np.float32)
    x64 = x.astype(np.float64)
    mean = x64.mean(1, keepdims=True)
    var = ((x64 - mean) ** 2).mean(1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + 1e-12)
    real = ((x64 - mean) * rstd * g + be) @ W.astype(np.float64).T + b
    xb = bf16_round(x)
    ident = rstd * (xb.astype(np.float64) @ Wf.astype(np.float64).T - mean * csum.astype(np.float64)) + bf
    if epi == EPI_BIAS_GELU:
        real, ident = _gelu(real), _gelu(ident)
    Mp = (M + 255) // 256 * 256
    dX = torch.from_numpy(x).cuda()
    dXb = torch.zeros((Mp, K), device="cuda", dtype=torch.bfloat16)
    dP = torch.zeros((Mp, 4, 2), device="cuda", dtype=torch.float32)
    eng.op_ln_prep(dX, dXb, dP, M)
    assert np.array_equal(dXb[:M].float().cpu().numpy(), xb)
    part = dP[:M].cpu().numpy().astype(np.float64)
    assert np.abs(part[:, 0, 0] - x64.sum(1)).max() <= 2e-6 * np.abs(x64).sum(1).max() and (part[:, 1:] == 0).all()
    # spread the sums over three slots, as the fp32-residual GEMM writes them (the consumer adds the four slots up)
    want = _ln_stats_ref(x64)
    dP[:M] = torch.from_numpy(want.astype(np.float32)).cuda()
    dW, dB, dC = _dev(Wf, "bf16"), torch.from_numpy(bf).cuda(), torch.from_numpy(csum).cuda()
    dO = torch.full((M + 3, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize()
    eng.op_gemm_ln(dXb, dW, dB, dO, None, M, N, K, epi, tile, dP, dC, None)
    got = dO.float().cpu().numpy().astype(np.float64)
    assert np.isnan(got[M:]).all(), "rows behind M were written"
    e_ident = np.abs(got[:M] - ident).max() / np.abs(ident).max()
    e_real = np.abs(got[:M] - real).max() / np.abs(real).max()
    report(f"persistent gemm with folded LN tile{tile} M{M} N{N} epi{epi}: vs identity {e_ident:.3e} (tol 6e-3), vs LN(x) W^T {e_real:.3e} (tol 2e-2)")
    assert np.isfinite(got[:M]).all() and e_ident <= 6e-3 and e_real <= 2e-2


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K,split", [(64, 768, 768, 12), (37, 2304, 768, 4), (128, 768, 3072, 16), (64, 6144, 768, 2)])
def test_gemm_split_k_slabs(dtype, M, N, K, split):
    eng = engine(dtype)
    rs = np.random.RandomState(split)
    Mp = (M + 63) // 64 * 64
    A = rs.standard_normal((Mp, K)).astype(np.float32)
    W = (rs.standard_normal((N, K)) * 0.05).astype(np.float32)
    if dtype == "bf16":
        A, W = bf16_round(A), bf16_round(W)
    dA, dW = _dev(A, dtype), _dev(W, dtype)
    dO = torch.full((split, M, N), float("nan"), device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    eng.op_gemm(dA, dW, None, dO, None, M, N, K, EPI_SLAB, tile=64, split_k=split)
    slabs = dO.cpu().numpy().astype(np.float64)
    ks = K // split
    worst = 0.0
    for z in range(split):
        ref = A[:M, z * ks:(z + 1) * ks].astype(np.float64) @ W[:, z * ks:(z + 1) * ks].astype(np.float64).T
        worst = max(worst, np.abs(slabs[z] - ref).max() / np.abs(ref).max())
    report(f"gemm split-K {dtype} M{M} N{N} K{K} split{split}: worst slab rel err {worst:.3e}")
    assert worst <= 2e-5


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_layernorm(dtype):
    eng = engine(dtype)
    rs = np.random.RandomState(5)
    M = 403
    x = (rs.standard_normal((M, 768)) * 3 + 0.7).astype(np.float32)
    g = (1 + 0.1 * rs.standard_normal(768)).astype(np.float32)
    b = (0.1 * rs.standard_normal(768)).astype(np.float32)
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x), (768,), torch.from_numpy(g), torch.from_numpy(b), 1e-12).numpy()
    dO = torch.full((M, 768), float("nan"), device="cuda", dtype=torch.float32 if dtype == "fp32" else torch.bfloat16)
    dx, dg, db = (torch.from_numpy(a).cuda() for a in (x, g, b))
    torch.cuda.synchronize()
    eng.op_layernorm(dx, dg, db, dO, M)
    got = dO.float().cpu().numpy()
    err = np.abs(got - ref).max()
    tol = 2e-5 if dtype == "fp32" else 2e-2
    report(f"layernorm {dtype}: max abs err {err:.3e}")
    assert err <= tol


@pytest.mark.parametrize("dtype,impl,n", [("fp32", 0, 3), ("fp32", 1, 3), ("fp32", 1, 1), ("fp32", 1, 30), ("bf16", 0, 3), ("bf16", 1, 3), ("bf16", 1, 1),
                                          ("bf16", 1, 30), ("bf16", 1, 70)] +
                         ([("bf16", 2, 3)] if LAB else []))
def test_encoder_attention(dtype, impl, n):
    """impl 0: the VALU kernel; 1: fp32 - the parity mode's kernel on the f32-input matrix cores (enc_attn_f32_kernel: exact fp32
    products), bf16 - the MFMA kernel of the bf16 engine (K / V by LDS-DMA, V read
    column-wise with ds_read_b64_tr_b16) - 1 crop (four blocks share a head's 13 query units), 3 / 30 (two), 70 (one block
    per head, three blocks per CU); 2: the r02 kernel (experiments build)."""
    eng = engine(dtype)
    rs = np.random.RandomState(11 + impl + n)
    S, H, dh = 197, 12, 64
    qkv = (rs.standard_normal((n * S, 3 * H * dh)) * 1.5).astype(np.float32)
    if dtype == "bf16":
        qkv = bf16_round(qkv)
    t = torch.from_numpy(qkv).view(n, S, 3, H, dh).permute(2, 0, 3, 1, 4).double()
    s = torch.matmul(t[0], t[1].transpose(-1, -2)) * 0.125
    ref = torch.matmul(torch.softmax(s, dim=-1), t[2]).permute(0, 2, 1, 3).reshape(n * S, H * dh).numpy()
    dQ = _dev(qkv, dtype)
    # rows behind n * S: no kernel may read them (r03 advisor: enc_attn2_kernel's K / V copy used to run 3 rows past the last image)
    pad = torch.full((256, 3 * H * dh), float("nan"), device="cuda", dtype=dQ.dtype)
    dQ = torch.cat([dQ, pad]).contiguous()
    dC = torch.full((n * S, H * dh), float("nan"), device="cuda", dtype=dQ.dtype)
    torch.cuda.synchronize()
    eng.op_enc_attention(dQ, dC, n, impl)
    got = dC.float().cpu().numpy().astype(np.float64)
    err = np.abs(got - ref).max()
    tol = 1e-5 if dtype == "fp32" else 2e-2      # fp32: 2e-6 of the largest value - a few hundred fp32 additions per output, in either order
    report(f"enc attention {dtype} impl{impl}: max abs err {err:.3e} (|ref| max {np.abs(ref).max():.2f})")
    assert np.isfinite(got).all()
    assert err <= tol


@pytest.mark.parametrize("tile", [16, 32])
@pytest.mark.parametrize("L,n", [(1, 5), (5, 5), (16, 5), (17, 5), (32, 5), (33, 5), (48, 5), (64, 5), (197, 5), (300, 5),
                                 (20, 300), (40, 300), (70, 700), (197, 520), (20, 1100), (40, 1300), (20, 2000), (197, 1600)])
def test_latent_attention(L, n, tile):
    """softmax(Qt X^T) X per head with the heads on the MFMA rows; keys streamed through the LDS ring
    in tiles of 16 keys on three persistent blocks per CU (the default kernel: transposed score tile, two-slot rings,
    kernels_latent_t.h) or of 32 keys (MOCR_FLAG_LATENT_TILE32: r03's kernel shape); L = 1 .. 300 covers 1 to 19 tiles, partial
    last tiles and the ring's drain; n > 768: persistent blocks take several sequences each (1, 2 and 3+ tile rows)."""
    eng = engine("bf16", flags=1024 if tile == 32 else 0)
    rs = np.random.RandomState(L)
    H, D = 12, 768
    stride = (L + 7) * D                                      # sequences are not tile-aligned in memory
    qt = np.zeros((n, 16, D), np.float32)
    qt[:, :H] = bf16_round((rs.standard_normal((n, H, D)) * 0.08).astype(np.float32))
    x = bf16_round(rs.standard_normal((n * (L + 7) + 64, D)).astype(np.float32))     # + rows a last tile may touch
    xs = np.stack([x[b * (L + 7): b * (L + 7) + L] for b in range(n)]).astype(np.float64)   # [n,L,D]
    sc = np.einsum("bhd,bkd->bhk", qt[:, :H].astype(np.float64), xs)
    sc -= sc.max(-1, keepdims=True)
    pr = np.exp(sc)
    pr /= pr.sum(-1, keepdims=True)
    ref = np.einsum("bhk,bkd->bhd", pr, xs)
    dq, dx = _dev(qt, "bf16"), _dev(x, "bf16")
    do = torch.full((n, 16, D), float("nan"), device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize()
    eng.op_latent_attention(dq, dx, do, n, L, stride)
    got = do[:, :H].float().cpu().numpy().astype(np.float64)
    err = np.abs(got - ref).max()
    report(f"latent attention tile {tile} L={L} n={n}: max abs err {err:.3e} (|ref| max {np.abs(ref).max():.2f})")
    assert np.isfinite(got).all() and err <= 3e-2


@pytest.mark.parametrize("n", [128, 300, 1024, 1408])
def test_fused_query_kernel(n):
    """dec_qqt_kernel: Qt[m][h] = bf16(x[m] Wq_h^T + bq_h) . wkT_h for the 12 heads, against float64 numpy with the same
    intermediate bf16 rounding of q.  Up to 1280 rows the launch runs on 64-row blocks (eight-slot ring), above on 128-row blocks."""
    eng = engine("bf16")
    rs = np.random.RandomState(n)
    D, H = 768, 12
    npad = (n + 127) // 128 * 128
    x = bf16_round(rs.standard_normal((npad, D)).astype(np.float32))
    wq = bf16_round((rs.standard_normal((D, D)) * 0.04).astype(np.float32))
    bq = (rs.standard_normal(D) * 0.1).astype(np.float32)
    wkT = bf16_round((rs.standard_normal((D, D)) * 0.04).astype(np.float32))      # [n][64h + k]
    q = bf16_round((x.astype(np.float64) @ wq.astype(np.float64).T + bq).astype(np.float32)).astype(np.float64)
    ref = np.einsum("mhk,nhk->mhn", q.reshape(npad, H, 64), wkT.astype(np.float64).reshape(D, H, 64))
    dqt = torch.full((npad, 16, D), float("nan"), device="cuda", dtype=torch.bfloat16)
    dx, dwq, dwk = _dev(x, "bf16"), _dev(wq, "bf16"), _dev(wkT, "bf16")
    dbq = torch.from_numpy(bq).cuda()
    torch.cuda.synchronize()
    eng.op_qqt(dx, dwq, dbq, dwk, dqt, n)
    got = dqt[:n, :H].float().cpu().numpy().astype(np.float64)
    err = np.abs(got - ref[:n]).max() / np.abs(ref).max()
    report(f"fused q->Qt kernel n={n}: max rel err {err:.3e}")
    assert np.isfinite(got).all() and err <= 8e-3
