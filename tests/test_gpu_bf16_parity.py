"""-m gpu: parity of the BENCHMARKED configuration - the bf16 engine on its automatic path (classic projected-K/V
kernels up to 256 rows, latent attention above), through the C ABI, at BASELINE configs[1] / configs[2] row counts,
max_len 300 - against the reference arithmetic.

What "bit-identical decoded token ids" can mean for a bf16 engine: greedy decoding is a chain of argmax decisions, and
a bf16 engine's logits differ from the fp32 reference's by up to ~1.3e-2 (measured below), so a decision whose top-2
margin IN THE REFERENCE is smaller than that can flip - after which the row legitimately continues differently.  The
tests therefore require: every row is identical to the reference up to its FIRST divergence, and that divergence sits at
a decision whose reference margin is below BF16_GAP_TOL; the teacher-forced logits stay within BF16_LOGIT_TOL of the
oracle.  The reference ids and margins come from tests/golden/bf16_parity.npz, written by tests/golden/make_goldens.py
from transformers' own generate() (greedy) - 256 crops x 300 tokens with plain synthetic weights (seed 0) and 32 crops
with widened-margin weights (seed 2, vocab_bias_std 2.0)."""
import os

import numpy as np
import pytest

from gpu_util import crops, engine, oracle, report

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

BF16_LOGIT_TOL = 3e-2     # max |logit - oracle logit| accepted, teacher-forced (measured: 1.2e-2 .. 1.4e-2)
BF16_GAP_TOL = 1.5e-2     # a first divergence is accepted only where the reference's own top-2 margin is below this (measured: <= 1.09e-2;
                          # r02 ran with 2.5e-2)


@pytest.fixture(scope="module")
def gold(golden_dir):
    g = np.load(os.path.join(golden_dir, "bf16_parity.npz"))
    return {k: g[k] for k in g.files}


def first_divergences(got, want, gaps, what):
    """Rows must equal the reference up to their first divergence; returns (rows identical, list of (row, token, gap))."""
    got, want = np.asarray(got), np.asarray(want).astype(np.int32)
    div = []
    for b in range(got.shape[0]):
        neq = np.nonzero(got[b] != want[b])[0]
        if neq.size:
            t = int(neq[0])
            div.append((b, t, float(gaps[b, t - 1])))      # ids[t] was decided at step t-1
    worst = max((g for _, _, g in div), default=0.0)
    n = got.shape[0]
    toks_before = sum(t for _, t, _ in div) + (n - len(div)) * got.shape[1]
    report(f"[{what}] {n} rows x {got.shape[1]} tokens: {n - len(div)} rows identical to the reference; {len(div)} rows diverge, first at "
           f"token {np.mean([t for _, t, _ in div]) if div else float('nan'):.0f} on average, reference top-2 margin there: "
           f"max {worst:.3e}, median {np.median([g for _, _, g in div]) if div else float('nan'):.3e}; "
           f"{toks_before / got.size * 100:.1f}% of all tokens precede any divergence")
    bad = [(b, t, g) for b, t, g in div if not g < BF16_GAP_TOL]
    assert not bad, f"{what}: rows diverge from the reference away from a numerical tie (row, token, reference margin): {bad[:5]}"
    return n - len(div), div


@pytest.mark.parametrize("rows", [64, 96, 128, 192, 256])
def test_bf16_auto_path_free_running_ids_against_the_reference(gold, rows):
    """BASELINE configs[1] (64 rows) and configs[2] (256 rows): the product's automatic kernel choice, max_len 300.  96, 128
    and 192 rows sit in the other encoder regimes (r03): one round of 256 x 256 tiles for the N = 768 GEMMs with the
    LayerNorms folded (56-111 crops), the 128 x 128 kernels for them with LayerNorm launches (112-163), two short rounds
    of the persistent kernel (164-221)."""
    eng = engine("bf16", max_batch=256, auto_path=True)
    gray = crops(777, 256)[:rows]
    ids, lens = eng.recognize(gray)
    assert ids.shape == (rows, 300) and (lens == 300).all() and (ids[:, 0] == 2).all()
    same, div = first_divergences(ids, gold["ids_seed0"][:rows], gold["gaps_seed0"].astype(np.float32), f"bf16 auto path, {rows} rows, plain weights")
    # the id-match rate over whole rows is reported, not asserted: with random weights ~2.5 % of all decisions have a
    # margin below the bf16 noise, so most 299-decision rows meet one


@pytest.mark.parametrize("rows,path", [(320, "latent"), (448, "latent")])
def test_bf16_fat_batch_free_running_ids(gold, rows, path):
    """Above configs[2] (256 rows, the last row count on the classic kernels): 320 and 448 rows take the latent attention, like
    the bench's merged batches.  The rows are the 256 golden crops + repeats."""
    eng = engine("bf16", max_batch=rows, auto_path=True)
    base = crops(777, 256)
    gray = np.concatenate([base, base[:rows - 256]])
    ids, _ = eng.recognize(gray)
    want = np.concatenate([gold["ids_seed0"], gold["ids_seed0"][:rows - 256]])
    gaps = np.concatenate([gold["gaps_seed0"], gold["gaps_seed0"][:rows - 256]]).astype(np.float32)
    first_divergences(ids, want, gaps, f"bf16 auto path, {rows} rows ({path} attention), plain weights")
    np.testing.assert_array_equal(ids[:rows - 256], ids[256:])          # a row does not depend on its position in the batch


@pytest.mark.parametrize("rows,lanes,max_batch", [(1024, 1, 1024), (2560, 2, 5120)])
def test_bf16_bench_shape_free_running_ids(gold, rows, lanes, max_batch):
    """The shapes bench.py's timed region runs: device-resident steps of 256 crops submitted back to back
    (mocr_recognize_device) and merged by the engine - 4 x 256 -> ONE 1024-row batch (128 x 128 decode tiles, the fused
    query kernel, wide encoder GEMMs), and 20 x 256 on a two-lane engine -> the queue split into 2 x 2560 rows (the
    bench's default: split-K target of >= 2048 rows, both lanes in flight; max_batch = the queue's 5120 rows, so nothing is
    pumped before the whole queue is in and the idle-lane split of pump_once decides the batch shapes, as in bench.py).  Every step is the 256 golden crops, so every
    row is held to the reference by the first-divergence rule and a crop must decode to the same ids whatever step,
    position or lane it was in."""
    from gpu_util import drop_engines
    drop_engines()                                  # fat workspaces: give the HBM of the cached engines back first
    eng = engine("bf16", max_batch=max_batch, auto_path=True, lanes=lanes)
    base = crops(777, 256)
    steps = rows // 256 * lanes
    dg = torch.from_numpy(base).cuda()
    d_ids = torch.zeros((steps, 256, 300), dtype=torch.int32, device="cuda")
    d_len = torch.zeros((steps, 256), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for i in range(steps):
        eng.recognize_device(dg, 256, d_ids[i], d_len[i])
    eng.synchronize()
    torch.cuda.synchronize()
    ids = d_ids.cpu().numpy()
    assert (d_len.cpu().numpy() == 300).all() and (ids[:, :, 0] == 2).all()
    gaps = gold["gaps_seed0"].astype(np.float32)
    first_divergences(ids[0], gold["ids_seed0"], gaps, f"bf16 bench shape, {steps} x 256 crops merged into {lanes} x {rows} rows, step 0")
    first_divergences(ids[-1], gold["ids_seed0"], gaps, f"bf16 bench shape, {steps} x 256 crops merged into {lanes} x {rows} rows, last step")
    for i in range(1, steps):
        np.testing.assert_array_equal(ids[i], ids[0], err_msg=f"step {i} of the merged queue decodes differently from step 0")
    drop_engines()


def test_bf16_2560_row_batch_teacher_forced_logits(gold):
    """One 2560-row internal batch (the row count of the bench's lanes), teacher-forced with the reference's ids for 24
    steps: rows 0..7 and 2552..2559 against the CPU oracle - the kernels that only switch on at fat batches (fused query
    kernel, 128 x 128 decode tiles, two-slab split-K, wide encoder GEMMs at M = 504,320) against the reference itself."""
    from gpu_util import drop_engines
    drop_engines()
    rows, T = 2560, 24
    eng = engine("bf16", max_batch=rows, auto_path=True)
    base = crops(777, 256)
    gray = np.concatenate([base] * (rows // 256))
    forced = np.concatenate([gold["ids_seed0"][:, :T]] * (rows // 256)).astype(np.int32)
    dg = torch.from_numpy(gray).cuda()
    torch.cuda.synchronize()
    got_all = eng.decode_logits(dg, rows, forced)
    sel = list(range(8)) + list(range(rows - 8, rows))
    got = got_all[sel]
    del got_all
    o = oracle()
    src = [r % 256 for r in sel]
    enc = o.encode(o.preprocess_gray(base[src]))
    _, ref = o.generate(enc, return_logits=True, forced_ids=gold["ids_seed0"][src, :T + 1].astype(np.int64))
    ref = ref[:, :T]
    d = np.abs(got - ref)
    srt = np.sort(ref, axis=-1)
    gap = srt[..., -1] - srt[..., -2]
    agree = got.argmax(-1) == ref.argmax(-1)
    report(f"teacher-forced logits bf16, ONE {rows}-row batch (rows 0..7 and {rows - 8}..{rows - 1} checked, {T} steps) vs oracle: max abs err "
           f"{d.max():.3e}, mean {d.mean():.3e}, argmax agreement {agree.mean():.4f}; disagreements all at reference margins < "
           f"{gap[~agree].max() if (~agree).any() else 0:.3e}")
    assert np.isfinite(got).all() and d.max() <= BF16_LOGIT_TOL
    assert (gap[~agree] < BF16_LOGIT_TOL).all(), "argmax differs where the reference's margin exceeds the logit tolerance"
    assert agree.mean() >= 0.97
    drop_engines()


@pytest.mark.parametrize("name,flags,auto", [("classic", 8, False), ("latent", 0, False), ("auto", 0, True)])
def test_bf16_widened_margin_weights_ids(gold, name, flags, auto):
    """Widened-margin weights (an fp32-exact N(0, 2^2) spread on the vocabulary bias): near-ties are ~3x rarer, so most
    rows must now be identical over all 300 tokens - on each of the three attention paths."""
    eng = engine("bf16", seed=2, vocab_bias_std=2.0, max_batch=32, flags=flags, auto_path=auto)
    ids, lens = eng.recognize(crops(778, 32))
    same, _ = first_divergences(ids, gold["ids_wide"], gold["gaps_wide"].astype(np.float32), f"bf16 {name} path, 32 rows, widened-margin weights")
    assert same >= 16, f"only {same} of 32 rows identical with widened margins"


@pytest.mark.parametrize("rows", [64, 256])
def test_bf16_auto_path_teacher_forced_logits(gold, rows):
    """The same batches, teacher-forced with the reference's ids for 47 steps; rows 0..7 against the CPU oracle."""
    eng = engine("bf16", max_batch=256, auto_path=True)
    gray = crops(777, 256)[:rows]
    forced = gold["ids_seed0"][:rows, :47].astype(np.int32)
    dg = torch.from_numpy(gray).cuda()
    torch.cuda.synchronize()
    got = eng.decode_logits(dg, rows, forced)[:8]
    o = oracle()
    enc = o.encode(o.preprocess_gray(gray[:8]))
    _, ref = o.generate(enc, return_logits=True, forced_ids=gold["ids_seed0"][:8, :48].astype(np.int64))
    ref = ref[:, :47]
    d = np.abs(got - ref)
    srt = np.sort(ref, axis=-1)
    gap = srt[..., -1] - srt[..., -2]
    agree = got.argmax(-1) == ref.argmax(-1)
    report(f"teacher-forced logits bf16 auto path ({rows} rows, rows 0..7 checked) vs oracle: max abs err {d.max():.3e}, mean {d.mean():.3e}, "
           f"argmax agreement {agree.mean():.4f}; disagreements all at reference margins < {gap[~agree].max() if (~agree).any() else 0:.3e}")
    assert np.isfinite(got).all() and d.max() <= BF16_LOGIT_TOL
    assert (gap[~agree] < BF16_LOGIT_TOL).all(), "argmax differs where the reference's margin exceeds the logit tolerance"
    assert agree.mean() >= 0.97


def test_bf16_encoder_with_folded_layernorm_against_separate_launches_and_the_reference():
    """From two 256 x 256 tiles per CU (~222 crops) the encoder's layer GEMMs run on the persistent kernel and the 24
    LayerNorms between them are folded into those GEMMs (A = the residual rows themselves as bf16, statistics applied in the
    epilogue; MOCR_FLAG_NO_LN_FOLD = 512 keeps them as launches).  Both forms against the fp32 oracle on the first crops, and
    against each other on all 256: the fold must not cost accuracy."""
    rows = 256
    gray = crops(777, rows)
    fold = engine("bf16", max_batch=rows)
    plain = engine("bf16", max_batch=rows, flags=512)
    dg = torch.from_numpy(gray).cuda()
    torch.cuda.synchronize()
    a = fold.encode(dg, rows).astype(np.float64)
    b = plain.encode(dg, rows).astype(np.float64)
    o = oracle()
    ref = o.encode(o.preprocess_gray(gray[:6])).numpy().astype(np.float64)
    scale = np.abs(ref).max()
    e_a = np.abs(a[:6] - ref).max() / scale
    e_b = np.abs(b[:6] - ref).max() / scale
    rms_a = np.sqrt(((a[:6] - ref) ** 2).mean()) / scale
    rms_b = np.sqrt(((b[:6] - ref) ** 2).mean()) / scale
    d_ab = np.abs(a - b).max() / scale
    report(f"bf16 encoder, {rows} crops: folded LayerNorm vs oracle max {e_a:.3e} rms {rms_a:.3e}; separate launches max {e_b:.3e} rms {rms_b:.3e}; "
           f"fold vs separate max {d_ab:.3e}")
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert not np.array_equal(a, b), "the flag did not change the path"
    assert e_a <= max(1.5 * e_b, 2e-2) and rms_a <= 1.3 * rms_b

