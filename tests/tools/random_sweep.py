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
    rng = np.random.default_rng(10_000 + seed)
    pairs = []
    while len(pairs) < 150:
        K, L = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        M, N = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        R = int(rng.choice([10, 12, 20, 30, 31]))
        band = str(rng.choice(["diag", "diag", "wander", "wander"]))
        A, B, LB, RB = inputs.make_pair(rng, K, L, M, N, R, band, mo.smooth,
                                        dash=float(rng.choice([0.0, 0.08, 0.35])), odd=float(rng.choice([0.0, 0.05, 0.6])))
        if mo.check(M, N, LB, RB)[0] == 0:
            pairs.append((A, B, LB, RB))
    batch = synth.pack_pairs(pairs)
    db = mz.DevBatch(batch); db.run(); res = db.results(); out = db.out.cpu().numpy()
    modes += np.bincount(res["mode"], minlength=14)
    om, hs, cells, nbad = mo.yama_batch(batch, variant=1, threads=8)
    for i in range(len(pairs)):
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        w = pairs[i][0].shape[1] + pairs[i][1].shape[1]
        ok = res["status"][i] == 0 and m_ == om[i] and mo.fnv1a_np(out[o0:o0 + m_ * w], mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8))) == int(hs[i])
        tot += 1
        if not ok:
            bad += 1
            print("BAD seed", seed, "pair", i, "mode", int(res["mode"][i]), "shape", pairs[i][0].shape, pairs[i][1].shape, flush=True)
print("pairs", tot, "bad", bad, "modes", modes)
