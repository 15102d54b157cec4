"""Differential sweep of the host-buffer path (mz_yama_batch: class nibbles and band steps up -- nibble, byte and raw
formats --, 2-bit scripts back, merged columns assembled on the host by shape-specific and generic code, chunks of every
size through the three-stage pipeline) against the oracle: random shapes incl. thin (shuffle assembly), mid (fixed-size
moves) and wide blocks (16-byte copies), bands that wander, jump by tens or hundreds of columns, or are full; pairs the
plan refuses in between.    python tests/tools/host_sweep.py <seed0> <seed1>     (MZ_CHUNK_PAIRS=<n> to vary the chunking)"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import inputs
import multiz_amd as mz
from oracle import mzoracle as mo
mz.api.init(0)
tot = bad = refused = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(910_000 + seed)
    pairs, want_bad = [], []
    while len(pairs) < 400:
        kind = int(rng.integers(0, 7))
        K, L = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        M, N = int(rng.integers(1, 600)), int(rng.integers(1, 600))
        R = int(rng.choice([3, 10, 30, 31, 64]))
        band = str(rng.choice(["diag", "wander", "wander", "full"]))
        if kind == 0:   K, L = int(rng.integers(5, 40)), int(rng.integers(5, 40)); M, N = int(rng.integers(5, 200)), int(rng.integers(5, 200))
        elif kind == 1: K, L = int(rng.integers(1, 3)), int(rng.integers(1, 3)); M, N = int(rng.integers(800, 3000)), int(rng.integers(800, 3000)); band = "diag"
        A, B, LB, RB = inputs.make_pair(rng, K, L, M, N, R, band, mo.smooth,
                                        dash=float(rng.choice([0.0, 0.08, 0.5])), odd=float(rng.choice([0.0, 0.05, 0.6])))
        if band == "full" and M * N > 60000:
            continue
        if kind == 2 and M > 8:                      # a jump of tens / hundreds of columns in the bounds (byte / raw step formats)
            j = int(rng.integers(2, M - 2)); jump = int(rng.choice([20, 90, 300, 700]))
            if N > jump + 40:
                LB = LB.copy(); RB = RB.copy()
                RB[j:] = np.minimum(RB[j:] + jump, N); LB[j + 1:] = np.minimum(LB[j + 1:] + jump, N - 11 if N > 11 else 0)
                LB = np.maximum.accumulate(LB); RB = np.maximum.accumulate(RB); RB[-1] = N
        rc = mo.check(M, N, LB, RB)[0]
        if rc != 0 and rng.random() < 0.9:
            continue
        pairs.append((A, B, LB.astype(np.int32), RB.astype(np.int32))); want_bad.append(rc)
    res = mz.yama_batch(pairs)
    for i, ((A, B, LB, RB), r) in enumerate(zip(pairs, res)):
        tot += 1
        if want_bad[i]:
            refused += 1
            if r.status != want_bad[i]:
                bad += 1; print("BAD status seed", seed, "pair", i, r.status, want_bad[i], flush=True)
            continue
        w = mo.yama(A, B, LB, RB, variant="profile")
        if not (r.status == 0 and r.OM == w.OM and np.array_equal(r.cols, w.cols)):
            bad += 1
            print("BAD seed", seed, "pair", i, "status", r.status, "shape", A.shape, B.shape, flush=True)
print("pairs", tot, "refused as the reference would", refused, "bad", bad)
