"""Bands as pre_yama derives them from blocks with indels against the shared reference row
(mz_preyama.c:240-258 then smooth()): the band centre stands still over columns only the first block has and
jumps over columns only the second block has.  Mode histogram, kernel times, every pair against the oracle.

    python tests/tools/indel_bands.py <pairs> <indel events per 1000 columns> [mean indel length] [nocheck]
"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import inputs
import multiz_amd as mz
from multiz_amd import synth
from oracle import mzoracle as mo

mz.api.init(0)
n = int(sys.argv[1]); rate = float(sys.argv[2]) / 1000.0
mean_len = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
rng = np.random.default_rng(11)
pairs = []
for p in range(n):
    M = int(rng.integers(900, 1101))
    centre = np.zeros(M + 1, dtype=np.int64)
    c = 0; i = 1
    while i <= M:
        u = rng.random()
        if u < rate / 2 and i > 1:                       # columns only block 1 has: the centre stands still
            g = int(rng.geometric(1.0 / mean_len))
            for _ in range(min(g, M - i + 1)):
                centre[i] = c; i += 1
            continue
        if u < rate:                                     # columns only block 2 has: the centre jumps
            c += int(rng.geometric(1.0 / mean_len))
        c += 1
        centre[i] = c; i += 1
    N = int(max(c, 11))
    LB = np.minimum(centre, N).astype(np.int32); RB = LB.copy(); LB[0] = 0; RB[M] = N
    LB, RB = mo.smooth(LB, RB, M, N, 30)
    A = inputs.random_block(rng, M, 2, dash=0.08, odd=0.05)
    B = inputs.noisy_copy(rng, A, N, 2, dash=0.08)
    pairs.append((A, B, LB, RB))
batch = synth.pack_pairs(pairs)
db = mz.DevBatch(batch); db.run(); res = db.results()
cells = int(res["cells"].sum())
ms = np.zeros(4)
for _ in range(3): ms += np.array(db.run(timed=True))
w = (batch["poolRB"].astype(np.int64) - batch["poolLB"] + 1)
print(f"{n} pairs, {sys.argv[2]} indel events / 1000 columns, mean length {mean_len}: widest row {int(w.max())}, mean width {w.mean():.1f}")
print("modes", np.bincount(res["mode"], minlength=12), "failed", int((res["status"] != 0).sum()),
      "kernel ms", np.round(ms / 3, 3), "GCUPS(dp)", round(cells / (ms[1] / 3 * 1e-3) / 1e9, 1), "GCUPS(serial)", round(cells / (ms.sum() / 3 * 1e-3) / 1e9, 1))
if "nocheck" in sys.argv: sys.exit(0)
om, hs, ccells, bad = mo.yama_batch(batch, variant=1, threads=64)
out = db.out.cpu().numpy(); mism = 0
for i in range(n):
    m_, o0 = int(res["om"][i]), int(res["offOut"][i])
    if m_ != om[i] or mo.fnv1a_np(out[o0:o0 + m_ * 4], mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8))) != int(hs[i]): mism += 1
print("mismatches", mism)
