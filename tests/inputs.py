"""Seeded random block-pair generators shared by the tests (numpy RNG; the benchmark's
xorshift64 generator of SURVEY.md section 8d lives in the product library, mz_synth)."""
from __future__ import annotations

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
ODD = np.frombuffer(b"acgtNnXR", dtype=np.uint8)


def random_block(rng, cols: int, rows: int, dash=0.08, odd=0.05) -> np.ndarray:
    """(cols, rows) uint8, column-major like the reference's A/B; every column has >=1 non-dash"""
    X = ACGT[rng.integers(0, 4, size=(cols, rows))]
    if odd > 0:
        m = rng.random((cols, rows)) < odd
        X = np.where(m, ODD[rng.integers(0, len(ODD), size=(cols, rows))], X)
    d = rng.random((cols, rows)) < dash
    X = np.where(d, np.uint8(ord("-")), X).astype(np.uint8)
    alld = (X == ord("-")).all(axis=1)
    if alld.any():
        X[alld, rng.integers(0, rows, size=int(alld.sum()))] = ACGT[rng.integers(0, 4, size=int(alld.sum()))]
    return np.ascontiguousarray(X)


def noisy_copy(rng, A: np.ndarray, N: int, L: int, sub=0.10, dash=0.08) -> np.ndarray:
    M, K = A.shape
    B = random_block(rng, N, L, dash=0.0, odd=0.0)
    n = min(M, N)
    src = A[:n, np.arange(L) % K]
    B[:n] = np.where(src == ord("-"), B[:n], src)
    s = rng.random((N, L)) < sub
    B = np.where(s, ACGT[rng.integers(0, 4, size=(N, L))], B)
    d = rng.random((N, L)) < dash
    B = np.where(d, np.uint8(ord("-")), B).astype(np.uint8)
    alld = (B == ord("-")).all(axis=1)
    if alld.any():
        B[alld, 0] = ord("A")
    return np.ascontiguousarray(B)


def diag_band(M: int, N: int):
    i = np.arange(M + 1, dtype=np.int64)
    LB = (i * N // M).astype(np.int32)
    RB = LB.copy()
    LB[0] = 0
    return LB, RB


def wander_band(rng, M: int, N: int, step=3):
    """a drifting, locally widening raw band (before smooth): random monotone path with
    occasional jumps, as an indel-rich shared reference row would produce"""
    pos = np.cumsum(rng.integers(0, step, size=M + 1)).astype(np.float64)
    pos = pos / max(pos[-1], 1) * N
    LB = np.floor(pos).astype(np.int32)
    RB = np.minimum(LB + rng.integers(0, 4, size=M + 1), N).astype(np.int32)
    LB[0] = 0
    RB[M] = N
    return LB, RB


def make_pair(rng, K, L, M, N, R=30, band="diag", smooth_fn=None, dash=0.08, odd=0.05):
    A = random_block(rng, M, K, dash=dash, odd=odd)
    B = noisy_copy(rng, A, N, L, dash=dash)
    if band == "diag":
        LB, RB = diag_band(M, N)
    elif band == "wander":
        LB, RB = wander_band(rng, M, N)
    elif band == "full":
        LB = np.zeros(M + 1, dtype=np.int32)
        RB = np.full(M + 1, N, dtype=np.int32)
    else:
        raise ValueError(band)
    if smooth_fn is not None and band != "full":
        LB, RB = smooth_fn(LB, RB, M, N, R)
    return A, B, LB, RB
