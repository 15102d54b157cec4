"""Differential sweep over the corners: many rows per block (K, L up to 100), one- and two-column blocks, radius
0..250 (full-matrix and strip-mined bands), indel-shaped bands, dash-heavy and odd-byte columns, unrelated
sequences.  Every pair against the oracle (profile variant, itself pinned to the reference), both with the plan's
own kernel choice and with the row-parallel / fast kernels switched off.
    python tests/tools/extreme_sweep.py <seed0> <seed1>"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import inputs
import multiz_amd as mz
from multiz_amd import synth
from oracle import mzoracle as mo
mz.api.init(0)
tot = bad = 0
modes = np.zeros(14, dtype=np.int64)
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(70_000 + seed)
    pairs = []
    while len(pairs) < 120:
        kind = int(rng.integers(0, 6))
        K, L = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        M, N = int(rng.integers(1, 500)), int(rng.integers(1, 500))
        R = int(rng.choice([0, 3, 10, 30, 31, 64, 100, 250]))
        band = str(rng.choice(["diag", "wander", "full"]))
        if kind == 0:   K, L = int(rng.integers(20, 101)), int(rng.integers(20, 101)); M, N = int(rng.integers(5, 120)), int(rng.integers(5, 120))
        elif kind == 1: M, N = int(rng.integers(1, 4)), int(rng.integers(1, 300))
        elif kind == 2: M, N = int(rng.integers(1, 300)), int(rng.integers(1, 4))
        elif kind == 3: M, N = int(rng.integers(600, 1500)), int(rng.integers(600, 1500)); R = int(rng.choice([30, 64, 100]))
        A, B, LB, RB = inputs.make_pair(rng, K, L, M, N, R, band, mo.smooth,
                                        dash=float(rng.choice([0.0, 0.08, 0.5])), odd=float(rng.choice([0.0, 0.05, 0.6])))
        if kind == 4:   B = inputs.random_block(rng, N, L, dash=0.2, odd=0.1)          # unrelated sequences
        if mo.check(M, N, LB, RB)[0] == 0:
            pairs.append((A, B, LB, RB))
    batch = synth.pack_pairs(pairs)
    om, hs, cells, nbad = mo.yama_batch(batch, variant=1, threads=16)
    for fast, row in ((1, 1), (1, 0), (0, 0)):
        mz.lib().mz_enable_fast(fast); mz.lib().mz_enable_row(row)
        db = mz.DevBatch(batch); db.run(); res = db.results(); out = db.out.cpu().numpy()
        modes += np.bincount(res["mode"], minlength=14)
        for i in range(len(pairs)):
            m_, o0 = int(res["om"][i]), int(res["offOut"][i])
            w = pairs[i][0].shape[1] + pairs[i][1].shape[1]
            ok = res["status"][i] == 0 and m_ == om[i] and mo.fnv1a_np(out[o0:o0 + m_ * w], mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8))) == int(hs[i])
            tot += 1
            if not ok:
                bad += 1
                print("BAD seed", seed, "pair", i, "kernels", (fast, row), "mode", int(res["mode"][i]), "status", int(res["status"][i]), "shape", pairs[i][0].shape, pairs[i][1].shape, flush=True)
print("pairs x kernel sets", tot, "bad", bad, "modes", modes)
