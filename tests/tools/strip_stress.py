"""Randomised parity run for the strip kernels (MZ_MODE_TSTRIP / MZ_MODE_STRIP): bands wide AND high -- large radii, long
indels, drifting wide bands, full matrices --, blocks of 1-6 rows, strip boundaries at 63 / 64 / 65 / 127 / 128 / 129 rows; every
pair against the oracle by hash.   python tests/tools/strip_stress.py <pairs> <seed> [bad.npz]     (STRIP_STRESS_ROWS=<n>: up to n rows a block)"""
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import inputs
import multiz_amd as mz
from multiz_amd import synth
from oracle import mzoracle as mo
mz.api.init(0)
n = int(sys.argv[1]); rng = np.random.default_rng(int(sys.argv[2]))
pairs = []
while len(pairs) < n:
    A, B, LB, RB = inputs.random_wide_pair(rng, max_rows=int(os.environ.get("STRIP_STRESS_ROWS", "6")))
    if mo.check(A.shape[0], B.shape[0], LB, RB)[0] == 0:
        pairs.append((A, B, LB, RB))
batch = synth.pack_pairs(pairs)
db = mz.DevBatch(batch); db.run(); res = db.results()
om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=min(64, os.cpu_count() or 8))
out = db.out.cpu().numpy(); mism = []; below = []
for i in range(n):
    W = pairs[i][0].shape[1] + pairs[i][1].shape[1]
    m_, o0 = int(res["om"][i]), int(res["offOut"][i])
    if res["status"][i] == 21:                               # MZ_E_SENTINEL: scores below the reference's MININT, where ITS walk leaves the band (include/mz_amd.h) -- reported, not reproduced
        K, L = pairs[i][0].shape[1], pairs[i][1].shape[1]
        assert K * L * 525 * (pairs[i][0].shape[0] + pairs[i][1].shape[0] + 2) >= 1 << 30, "MZ_E_SENTINEL on a pair whose scores cannot get there"
        below.append(i)
        continue
    if res["status"][i] != 0 or m_ != om[i] or mo.fnv1a_np(out[o0:o0 + m_ * W], mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8))) != int(hs[i]):
        mism.append((i, int(res["mode"][i]), int(res["status"][i]), pairs[i][0].shape, pairs[i][1].shape))
print("modes", np.bincount(res["mode"], minlength=14), "oracle-invalid", bad, "below the reference's sentinel (status 21)", len(below), "mismatches", len(mism), mism[:10])
if below and len(sys.argv) > 3:                             # (the pairs below the sentinel, smallest first: tests/golden/below_sentinel.npz is one of them)
    sv = {}
    for j, i in enumerate(sorted(below, key=lambda i: pairs[i][0].size + pairs[i][1].size)[:12]):
        A, B, LB, RB = pairs[i]
        sv[f"A{j}"] = A; sv[f"B{j}"] = B; sv[f"LB{j}"] = LB; sv[f"RB{j}"] = RB
    np.savez_compressed(sys.argv[3].replace(".npz", "_below.npz"), **sv)
if mism and len(sys.argv) > 3:
    sv = {}
    for j, (i, *_ ) in enumerate(mism[:20]):
        A, B, LB, RB = pairs[i]
        sv[f"A{j}"] = A; sv[f"B{j}"] = B; sv[f"LB{j}"] = LB; sv[f"RB{j}"] = RB
    np.savez_compressed(sys.argv[3], **sv)
sys.exit(1 if mism else 0)
