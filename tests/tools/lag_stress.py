"""Randomised parity run for the lagged kernel: pairs of 1-6 rows per block, radii 10-30, long indels, short and long
pairs; every pair against the oracle by hash.   python tests/tools/lag_stress.py <pairs> <seed>"""
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import inputs
import multiz_amd as mz
from multiz_amd import synth
from oracle import mzoracle as mo
mz.api.init(0)
n = int(sys.argv[1]); rng = np.random.default_rng(int(sys.argv[2]))
max_rows = int(os.environ.get("LAG_STRESS_ROWS", "6"))
pairs = [inputs.random_indel_pair(rng, max_rows) for _ in range(n)]
for _ in range(n // 3):                                   # and bands no aligner would produce
    M = int(rng.integers(40, 400))
    LB, RB, N = inputs.random_walk_band(rng, M)
    A = inputs.random_block(rng, M, int(rng.integers(1, 5)), dash=0.1, odd=0.05)
    pairs.append((A, inputs.noisy_copy(rng, A, N, int(rng.integers(1, 5)), dash=0.1), LB, RB))
n = len(pairs)
batch = synth.pack_pairs(pairs)
db = mz.DevBatch(batch); db.run(); res = db.results()
om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=min(64, os.cpu_count() or 8))
out = db.out.cpu().numpy(); mism = []
for i in range(n):
    W = pairs[i][0].shape[1] + pairs[i][1].shape[1]
    m_, o0 = int(res["om"][i]), int(res["offOut"][i])
    if res["status"][i] != 0 or m_ != om[i] or mo.fnv1a_np(out[o0:o0 + m_ * W], mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8))) != int(hs[i]):
        mism.append((i, int(res["mode"][i]), int(res["status"][i])))
print("modes", np.bincount(res["mode"], minlength=12), "oracle-invalid", bad, "mismatches", len(mism), mism[:10])
if mism and len(sys.argv) > 3:                      # save the pairs that differ (analysis off the GPU box)
    sv = {}
    for j, (i, _, _) in enumerate(mism[:20]):
        A, B, LB, RB = pairs[i]
        sv[f"A{j}"] = A; sv[f"B{j}"] = B; sv[f"LB{j}"] = LB; sv[f"RB{j}"] = RB
        w = mo.yama(A, B, LB, RB, variant="profile")
        m_, o0 = int(res["om"][i]), int(res["offOut"][i]); W = A.shape[1] + B.shape[1]
        sv[f"got{j}"] = out[o0:o0 + m_ * W].reshape(m_, W); sv[f"want{j}"] = w.cols
        sv[f"f3{j}"] = res["final3"][i]
    np.savez_compressed(sys.argv[3], **sv)
