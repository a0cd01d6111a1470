"""Rows that finish early must stop costing - and must not change anyone's ids (r04).

In the reference every crop is its own ``generate()`` call: a 6-character bubble stops at its own EOS
(``TF/generation/utils.py:2929-2937``, called per crop at ``src/ui/main_window.py:9801``, one job per pop at
``src/core/workers.py:213-225``).  The engine decodes merged batches in lockstep; between two chunks of steps it moves
the unfinished rows to the first decode slots and runs the following steps on fewer slots (``engine.hip: compact_rows``).
Checked here, through the C ABI:

* fp32 parity mode: the compacted batch's ids are the transformers golden ids (``tests/golden/early_eos_seed1.npz``) for
  every copy of the six golden crops, and bit-identical to an uncompacted run (``MOCR_FLAG_NO_COMPACTION``);
* bf16, every attention path (the latent kernel and its A/B partner, classic, fp8): compacted == uncompacted, ids and lengths,
  bit for bit - the batch keeps the kernel regime it started with, so no summation order changes;
* the compactions really happen (``mocr_compaction_count``), on one lane and with two lanes in flight.
"""
import os

import numpy as np
import pytest

from gpu_util import crops, engine, report

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")      # before the engine's library: one HIP runtime per process, torch's

NO_COMPACTION = 2048
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _mixed_crops(n, seed=4321):
    """n crops whose rows finish at very different steps with the EOS-biased weights (seed 1, eos_bias 1.1): the six
    golden crops (31 .. 106 tokens) first, then seeded random ones."""
    g = crops(seed, 6)
    if n <= 6:
        return g[:n]
    return np.concatenate([g, crops(seed + 1, n - 6)])


def test_fp32_compacted_batch_gives_the_golden_early_eos_ids():
    want = np.load(os.path.join(GOLD, "early_eos_seed1.npz"))["ids"]
    L = want.shape[1]
    gray = np.tile(crops(4321, 6), (16, 1, 1))                     # 96 rows: every golden crop 16 times
    eng = engine("fp32", seed=1, eos_bias=1.1, max_batch=96)
    base = eng.compaction_count()
    ids, lens = eng.recognize(gray)
    n_comp = eng.compaction_count() - base
    ref = engine("fp32", seed=1, eos_bias=1.1, max_batch=96, flags=NO_COMPACTION)
    ids0, lens0 = ref.recognize(gray)
    assert ref.compaction_count() == 0
    np.testing.assert_array_equal(ids, ids0)
    np.testing.assert_array_equal(lens, lens0)
    for k in range(16):
        np.testing.assert_array_equal(ids[6 * k:6 * k + 6, :L], want)
        assert (ids[6 * k:6 * k + 6, L:] == 0).all()
    report(f"fp32, 96 rows (16 x the six golden early-EOS crops, lens {sorted(set(lens.tolist()))}): {n_comp} compactions, "
           "ids identical to the transformers goldens and to the uncompacted run")
    assert n_comp >= 2


@pytest.mark.parametrize("name,rows,flags", [
    ("latent", 640, 64),
    ("latent, r03 kernel shape (32-key tiles)", 640, 64 | 1024),
    ("latent, fused query kernel", 1280, 64),
    ("classic (automatic choice at 96 rows)", 96, 0),
    ("fp8 attention", 640, 64 | 128),
])
def test_bf16_compacted_equals_uncompacted(name, rows, flags):
    gray = _mixed_crops(rows)
    auto = not (flags & 64)
    eng = engine("bf16", seed=1, eos_bias=1.1, max_batch=rows, flags=flags, auto_path=auto)
    base = eng.compaction_count()
    ids, lens = eng.recognize(gray)
    n_comp = eng.compaction_count() - base
    ref = engine("bf16", seed=1, eos_bias=1.1, max_batch=rows, flags=flags | NO_COMPACTION, auto_path=auto)
    ids0, lens0 = ref.recognize(gray)
    assert ref.compaction_count() == 0
    np.testing.assert_array_equal(lens, lens0)
    np.testing.assert_array_equal(ids, ids0)
    # finished rows are padded, unfinished ones ran to max_len
    for b in range(rows):
        assert (ids[b, lens[b]:] == 0).all()
    report(f"bf16 {name}, {rows} rows, lengths {int(lens.min())}..{int(lens.max())} (mean {lens.mean():.1f}): {n_comp} compactions, "
           "ids and lengths bit-identical to the uncompacted run")
    assert n_comp >= 2 and lens.min() < lens.max()


def test_two_lanes_compact_independently():
    """Two merged batches in flight, each compacting at its own pace; rows return in submission order."""
    rows = 1280
    gray = _mixed_crops(rows)
    eng = engine("bf16", seed=1, eos_bias=1.1, max_batch=640, flags=64, lanes=2)
    ref = engine("bf16", seed=1, eos_bias=1.1, max_batch=640, flags=64 | NO_COMPACTION, lanes=1)
    d = torch.from_numpy(gray).cuda()
    out = torch.zeros((rows, 300), dtype=torch.int32, device="cuda")
    ln = torch.zeros((rows,), dtype=torch.int32, device="cuda")
    base = eng.compaction_count()
    for i in range(0, rows, 128):                                   # ten jobs, merged by the engine into 2 x 640 rows
        eng.recognize_device(d[i:i + 128], 128, out[i:i + 128], ln[i:i + 128])
    eng.synchronize()
    ids0, lens0 = ref.recognize(gray)
    np.testing.assert_array_equal(out.cpu().numpy(), ids0)
    np.testing.assert_array_equal(ln.cpu().numpy(), lens0)
    assert eng.compaction_count() - base >= 4
