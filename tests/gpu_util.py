"""Helpers shared by the -m gpu tests: engines are expensive to build, so they are cached."""
import functools
import os

import numpy as np

from manga_ocr.weights import DEFAULT_SPEC, synthetic_weights

REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.txt")


def report(line: str) -> None:
    print(line, flush=True)
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def crops(seed, n):
    return np.random.RandomState(seed).randint(0, 256, size=(n, 224, 224), dtype=np.uint8)


@functools.lru_cache(maxsize=None)
def weights(seed=0, eos_bias=0.0, vocab_bias_std=0.0, hostile=False):
    return synthetic_weights(seed, eos_bias=eos_bias, vocab_bias_std=vocab_bias_std, hostile=hostile)


@functools.lru_cache(maxsize=None)
def engine(dtype="fp32", seed=0, eos_bias=0.0, max_batch=8, flags=0, lanes=1, auto_path=False, vocab_bias_std=0.0, hostile=False):
    """bf16 engines of the tests run the latent attention at every batch size (flag 64) unless they ask for the classic
    kernels (flag 8) or for the product's automatic choice (auto_path: classic up to 256 rows)."""
    from manga_ocr.engine import Engine
    if dtype == "bf16" and not auto_path and not (flags & 8):
        flags |= 64
    return Engine(weights(seed, eos_bias, vocab_bias_std, hostile), DEFAULT_SPEC, dtype=dtype, device=0, max_batch=max_batch, flags=flags, lanes=lanes)


def drop_engines():
    """Engines are cached for speed; a test that needs the HBM back (fat default-sized engines) closes them all."""
    engine.cache_clear()
    import gc
    gc.collect()


@functools.lru_cache(maxsize=None)
def oracle(seed=0, eos_bias=0.0, vocab_bias_std=0.0, hostile=False):
    from oracle.mocr_oracle import Oracle
    return Oracle(weights(seed, eos_bias, vocab_bias_std, hostile), DEFAULT_SPEC)


def bf16_round(a: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def assert_ids_match_up_to_ties(got, want, gap_fn, tol, what):
    """Token ids must be identical; the only divergence tolerated is at a step where the ORACLE's
    own top-2 logit gap is below `tol` (a numerical tie no fp32 implementation can be held to).
    gap_fn(row, step) -> oracle top-2 gap at that decision."""
    got, want = np.asarray(got), np.asarray(want)
    L = min(got.shape[1], want.shape[1])
    n_tie = 0
    for b in range(got.shape[0]):
        neq = np.nonzero(got[b, :L] != want[b, :L])[0]
        if neq.size == 0:
            continue
        t = int(neq[0])          # ids[t] was decided at step t-1
        gap = float(gap_fn(b, t - 1))
        report(f"[{what}] row {b}: first divergence at token {t} (oracle top-2 gap {gap:.3e})")
        assert gap < tol, f"{what}: row {b} diverges at token {t} where the oracle's margin is {gap:.3e} >= {tol}"
        n_tie += 1
    return n_tie
