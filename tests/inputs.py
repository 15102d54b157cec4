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


# ------------------------------------------------------------------ MAF block pairs for pre_yama

def _mutate(rng, base: int) -> int:
    return int(ACGT[rng.integers(0, 4)]) if rng.random() < 0.12 else base


def random_maf_block(rng, ref: np.ndarray, s: int, e: int, nrows: int, tag: str, pins=0.04, pdel=0.05, lower=0.03):
    """a block whose top row is ref[s:e] on 'ref.chr1' plus nrows-1 noisy species rows"""
    from oracle.mzoracle import Block, Row
    cols = []                                    # list of per-row bytes
    for p in range(s, e):
        if p > s and rng.random() < pins:        # columns where the reference row has a dash
            for _ in range(int(rng.integers(1, 4))):
                col = [DASHB] + [int(ACGT[rng.integers(0, 4)]) if rng.random() < 0.7 else DASHB for _ in range(nrows - 1)]
                if all(c == DASHB for c in col[1:]) and nrows > 1:
                    col[1 + int(rng.integers(0, nrows - 1))] = int(ACGT[rng.integers(0, 4)])
                if nrows > 1:
                    cols.append(col)
        col = [int(ref[p])] + [DASHB if rng.random() < pdel else _mutate(rng, int(ref[p])) for _ in range(nrows - 1)]
        cols.append(col)
    arr = np.array(cols, dtype=np.uint8)          # (columns, rows)
    if lower > 0:
        m = (rng.random(arr.shape) < lower) & (arr != DASHB)
        m[:, 0] = False                           # keep the shared reference row byte-identical in both blocks
        arr = np.where(m, arr | 0x20, arr).astype(np.uint8)
    rows = []
    for r in range(nrows):
        text = bytes(arr[:, r]).decode("ascii")
        size = sum(ch != "-" for ch in text)
        if r == 0:
            rows.append(Row(src="ref.chr1", start=s, size=size, strand="+", srcSize=len(ref) + 1000, text=text))
        else:
            if size == 0:                         # a MAF row must hold at least one base
                text = "A" + text[1:] if text[0] == "-" else text
                arr[0, r] = ord(text[0]); size = sum(ch != "-" for ch in text)
            st = int(rng.integers(0, 5000))
            rows.append(Row(src=f"{tag}{r}.chr{r}", start=st, size=size, strand="+-"[int(rng.integers(0, 2))], srcSize=st + size + 77, text=text))
    return Block(rows=rows)


DASHB = ord("-")


def random_block_pair(rng, n1: int, n2: int, length: int = 150):
    """two blocks sharing part of their reference row; returns (a1, a2, beg, end)"""
    ref = ACGT[rng.integers(0, 4, size=length + 200)]
    s1, s2 = int(rng.integers(0, 60)), int(rng.integers(0, 60))
    e1, e2 = s1 + int(rng.integers(length // 2, length)), s2 + int(rng.integers(length // 2, length))
    a1 = random_maf_block(rng, ref, s1, e1, n1, "x")
    a2 = random_maf_block(rng, ref, s2, e2, n2, "y")
    beg, end = max(s1, s2), min(e1, e2) - 1
    return a1, a2, beg, end


def random_maf_file(rng, ref: np.ndarray, nblocks: int, nrows: int, tag: str, stride: int = 260, blen=(120, 250)):
    """single-coverage list of blocks along ref (what multiz expects of each input file)"""
    blocks = []
    for k in range(nblocks):
        s = k * stride + int(rng.integers(0, 40))
        e = min(s + int(rng.integers(*blen)), (k + 1) * stride - 1, len(ref))
        if e - s < 30:
            continue
        blocks.append(random_maf_block(rng, ref, s, e, nrows, tag))
    return blocks


def write_maf(path: str, blocks) -> None:
    from oracle.mzoracle import format_block, score_range
    with open(path, "w") as f:
        f.write("##maf version=1 scoring=blastz\n")
        for b in blocks:
            b.score = score_range(b, 0, b.textSize)
            f.write(format_block(b))
        f.write("##eof maf\n")


def indel_band(rng, M, rate, mean_len, radius):
    """the band pre_yama derives for blocks with indels against the shared reference row (mz_preyama.c:240-258, then
    smooth): the centre stands still over columns only the first block has and jumps over columns only the second has"""
    from oracle import mzoracle as mo
    centre = np.zeros(M + 1, dtype=np.int64)
    c, i = 0, 1
    while i <= M:
        u = rng.random()
        if u < rate / 2 and i > 1:
            for _ in range(min(int(rng.geometric(1.0 / mean_len)), M - i + 1)):
                centre[i] = c; i += 1
            continue
        if u < rate:
            c += int(rng.geometric(1.0 / mean_len))
        c += 1
        centre[i] = c; i += 1
    N = int(max(c, 11))
    LB = np.minimum(centre, N).astype(np.int32); RB = LB.copy(); LB[0] = 0; RB[M] = N
    LB, RB = mo.smooth(LB, RB, M, N, radius)
    return LB, RB, N


def random_indel_pair(rng, max_rows=6):
    """a block pair of 1-max_rows rows each, 30-900 columns, indels of mean length 2-12 at 5-100 events per 1 000 columns,
    radius 10-30, up to 40 % dashes"""
    M = int(rng.integers(30, 900))
    LB, RB, N = indel_band(rng, M, rng.choice([0.005, 0.02, 0.05, 0.1]), rng.choice([2.0, 5.0, 12.0]), int(rng.integers(10, 31)))
    K, L = int(rng.integers(1, max_rows + 1)), int(rng.integers(1, max_rows + 1))
    A = random_block(rng, M, K, dash=float(rng.choice([0.0, 0.1, 0.4])), odd=0.05)
    return A, noisy_copy(rng, A, N, L, dash=float(rng.choice([0.0, 0.1, 0.4]))), LB, RB


def random_walk_band(rng, M):
    """a valid band that no aligner would produce: both edges random monotone walks with occasional jumps, rows 11 to
    ~120 columns wide, every row touching the one before"""
    LB = np.zeros(M + 1, dtype=np.int64); RB = np.zeros(M + 1, dtype=np.int64)
    lo, hi = 0, int(rng.integers(11, 64))
    for r in range(M + 1):
        LB[r], RB[r] = lo, hi
        prev_hi = hi
        hi += int(rng.choice([0, 1, 1, 1, 2, 3, int(rng.integers(0, 40))], p=[0.15, 0.3, 0.2, 0.1, 0.1, 0.1, 0.05]))
        lo += int(rng.choice([0, 1, 1, 1, 2, 3, int(rng.integers(0, 40))], p=[0.15, 0.3, 0.2, 0.1, 0.1, 0.1, 0.05]))
        lo = min(lo, prev_hi, hi - 11)                   # touches the row before, at least 11 columns wide
        lo = max(lo, int(LB[r]))
        if hi - lo > 120: lo = hi - 120
    N = int(RB[M])
    return LB.astype(np.int32), RB.astype(np.int32), N


def long_indel_band(rng, M, rate, radius, long_frac=0.02, short_mean=3.0, long_lo=70, long_hi=300):
    """indel_band() with heavy-tailed run lengths: most runs geometric with mean `short_mean`, a fraction `long_frac` of them
    uniform in [long_lo, long_hi] columns -- one such run makes the rows around it too wide for the row-parallel kernels and the
    anti-diagonals too high for the rolling wavefront"""
    from oracle import mzoracle as mo

    def run():
        return int(rng.integers(long_lo, long_hi + 1)) if rng.random() < long_frac else int(rng.geometric(1.0 / short_mean))
    centre = np.zeros(M + 1, dtype=np.int64)
    c, i = 0, 1
    while i <= M:
        u = rng.random()
        if u < rate / 2 and i > 1:
            for _ in range(min(run(), M - i + 1)):
                centre[i] = c; i += 1
            continue
        if u < rate:
            c += run()
        c += 1
        centre[i] = c; i += 1
    N = int(max(c, 11))
    LB = np.minimum(centre, N).astype(np.int32); RB = LB.copy(); LB[0] = 0; RB[M] = N
    LB, RB = mo.smooth(LB, RB, M, N, radius)
    return LB, RB, N


def random_wide_pair(rng, max_rows=6):
    """a block pair whose band is wide AND high: a large radius around the diagonal, a long indel, a drifting wide band, or the
    full matrix; 1-max_rows rows a block, 20-900 columns"""
    from oracle import mzoracle as mo
    kind = rng.choice(["radius", "radius", "indel", "indel", "wander", "full"])
    K, L = int(rng.integers(1, max_rows + 1)), int(rng.integers(1, max_rows + 1))
    if kind == "indel":
        M = int(rng.integers(100, 900))
        LB, RB, N = long_indel_band(rng, M, float(rng.choice([0.01, 0.03])), int(rng.integers(10, 40)), long_frac=float(rng.choice([0.1, 0.4])))
    else:
        M = int(rng.choice([int(rng.integers(20, 900)), 63, 64, 65, 127, 128, 129]))
        N = max(11, M + int(rng.integers(-M // 3, M // 3 + 1)))
        if kind == "full" and M * N > 60000:
            kind = "radius"
        if kind == "radius":
            LB, RB = diag_band(M, N)
            LB, RB = mo.smooth(LB, RB, M, N, int(rng.integers(64, 160)))
        elif kind == "wander":
            LB, RB = wander_band(rng, M, N, step=int(rng.integers(2, 6)))
            LB, RB = mo.smooth(LB, RB, M, N, int(rng.integers(40, 120)))
        else:
            LB = np.zeros(M + 1, dtype=np.int32); RB = np.full(M + 1, N, dtype=np.int32)
    A = random_block(rng, M, K, dash=float(rng.choice([0.0, 0.1, 0.4])), odd=0.05)
    return A, noisy_copy(rng, A, N, L, dash=float(rng.choice([0.0, 0.1, 0.4]))), LB, RB
