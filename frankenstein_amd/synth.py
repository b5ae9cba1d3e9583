"""Deterministic synthetic inputs / targets / weights for parity tests and the bench.

Everything is produced by ``numpy.random.default_rng(seed)`` in float32 so the same
bytes can be regenerated on the GPU box, in the oracle, and when the golden fixtures
are produced from the reference in the build container (SURVEY.md §8d "Synthetic
inputs").  Nothing here depends on the reference.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping, Sequence, Tuple

import numpy as np

SEED_INPUT = 1234
SEED_TOKENS = 1235
SEED_WEIGHTS = 42


def make_inputs(B: int, T: int, C: int = 256, seed: int = SEED_INPUT) -> np.ndarray:
    """x ~ N(0,1) float32 [B, T, C] (neural feature frames, no all-zero frames)."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((B, T, C), dtype=np.float32)


def make_tokens(B: int, L: int = 25, vocab: int = 50257, seed: int = SEED_TOKENS,
                eot: int = 50256) -> np.ndarray:
    """Token targets int64 [B, L]: first token = eot, uniform ids, a random-length -100
    tail of 1..10 positions (mirrors utils/data_utils.py:270-286 padding with -100)."""
    rng = np.random.default_rng(seed)
    tok = rng.integers(0, vocab, size=(B, L), dtype=np.int64)
    tok[:, 0] = min(eot, vocab - 1)
    tails = rng.integers(1, min(10, L - 2) + 1, size=(B,))
    for b in range(B):
        tok[b, L - int(tails[b]):] = -100
    return tok


def make_motion_targets(B: int, M: int, D: int, seed: int = SEED_TOKENS) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.standard_normal((B, M, D), dtype=np.float32)


def _key_seed(seed: int, name: str) -> int:
    return (seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF


def make_tensor(name: str, shape: Sequence[int], seed: int = SEED_WEIGHTS) -> np.ndarray:
    """One weight tensor, chosen by state-dict key name + rank.

    * norm weights  (``ln_*.weight``, 1-D ``weight``) : 1 + 0.1 N(0,1)
    * biases / 1-D                                  : 0.1 N(0,1)
    * 2-D Linear / Embedding                        : N(0, 1/sqrt(fan_in))
    * 3-D (space_embedding, learnable_queries)      : 0.5 N(0,1)
    * 3-D convolution ``weight`` [.., .., K]         : N(0, 1/sqrt(shape[1]*shape[2]))
    Activations stay O(1) so softmax / norms are exercised non-trivially.
    """
    rng = np.random.default_rng(_key_seed(seed, name))
    shape = tuple(int(s) for s in shape)
    z = rng.standard_normal(shape, dtype=np.float32)
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) == 1:
        if leaf == "weight":
            return (1.0 + 0.1 * z).astype(np.float32)
        return (0.1 * z).astype(np.float32)
    if len(shape) == 2:
        return (z / np.sqrt(np.float32(shape[1]))).astype(np.float32)
    if len(shape) == 3 and leaf == "weight":
        return (z / np.sqrt(np.float32(shape[1] * shape[2]))).astype(np.float32)
    return (0.5 * z).astype(np.float32)


def make_state(shapes: Mapping[str, Sequence[int]], seed: int = SEED_WEIGHTS,
               skip: Tuple[str, ...] = ("attn_mask",)) -> Dict[str, np.ndarray]:
    """Weights for every key of ``shapes`` (name -> shape); buffers in ``skip`` are left out."""
    out = {}
    for k, shp in shapes.items():
        if any(k.endswith(s) for s in skip):
            continue
        out[k] = make_tensor(k, shp, seed)
    return out
