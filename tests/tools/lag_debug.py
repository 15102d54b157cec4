"""Find pairs the lagged kernel gets wrong and save their bands (gpurun_out/lag_bad.npz) for analysis off the GPU box."""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import inputs
import multiz_amd as mz
from multiz_amd import synth
from oracle import mzoracle as mo
from test_gpu_parity import _indel_band_pair
mz.api.init(0)
n = int(sys.argv[1]); ev = float(sys.argv[2]); seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rng = np.random.default_rng(seed)
pairs = [_indel_band_pair(rng, ev / 1000.0) for _ in range(n)]
batch = synth.pack_pairs(pairs)
db = mz.DevBatch(batch); db.run(); res = db.results()
om, hs, ccells, bad = mo.yama_batch(batch, variant=1, threads=64)
out = db.out.cpu().numpy()
badl = []
for i in range(n):
    m_, o0 = int(res["om"][i]), int(res["offOut"][i])
    okk = res["status"][i] == 0 and m_ == om[i] and mo.fnv1a_np(out[o0:o0 + m_ * 4], mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8))) == int(hs[i])
    if not okk: badl.append(i)
print("modes", np.bincount(res["mode"], minlength=12), "bad", len(badl), [(i, int(res["mode"][i]), int(res["status"][i])) for i in badl[:20]])
sv = {}
for j, i in enumerate(badl[:40]):
    A, B, LB, RB = pairs[i]
    sv[f"LB{j}"] = LB; sv[f"RB{j}"] = RB; sv[f"A{j}"] = A; sv[f"B{j}"] = B
good = [i for i in range(n) if i not in set(badl) and res["mode"][i] == 11][:10]
for j, i in enumerate(good):
    sv[f"gLB{j}"] = pairs[i][2]; sv[f"gRB{j}"] = pairs[i][3]
np.savez_compressed("gpurun_out/lag_bad.npz", **sv)
# each bad pair alone: does it fail on its own?
for i in badl[:8]:
    b1 = synth.pack_pairs([pairs[i]])
    d1 = mz.DevBatch(b1); d1.run(); r1 = d1.results()
    w = mo.yama(*pairs[i], variant="profile")
    o1 = d1.out.cpu().numpy()
    same = r1["status"][0] == 0 and int(r1["om"][0]) == w.OM and np.array_equal(o1[:w.OM * 4].reshape(w.OM, 4), w.cols)
    print("pair", i, "alone: status", int(r1["status"][0]), "mode", int(r1["mode"][0]), "same", bool(same), "score", int(r1["score"][0]) if "score" in r1 else None, w.score if hasattr(w, "score") else None)
