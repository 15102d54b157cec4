import sys, numpy as np, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import inputs
import multiz_amd as mz
from multiz_amd import synth
from oracle import mzoracle as mo
mz.api.init(0)
rng = np.random.default_rng(4)
n = int(sys.argv[1])
batches = []
for g in range(n // 250):
    K, L = int(rng.integers(1, 30)), int(rng.integers(1, 30))
    batches.append(synth.make_batch(250, K, L, 200, 1000, 30, first_pair=1000 * g))
# merge batches
def merge(bs):
    out = {k: np.concatenate([b[k] for b in bs]) for k in ("K", "L", "M", "N", "poolA", "poolB", "poolLB", "poolRB")}
    for key, pool in (("offA", "poolA"), ("offB", "poolB"), ("offBand", "poolLB")):
        off = 0; parts = []
        for b in bs:
            parts.append(b[key] + off); off += len(b[pool])
        out[key] = np.concatenate(parts)
    return out
batch = merge(batches)
db = mz.DevBatch(batch); db.run(); res = db.results()
cells = int(res["cells"].sum())
ms = np.zeros(4)
for _ in range(3): ms += np.array(db.run(timed=True))
print("C4-style", len(batch["K"]), "pairs; modes", np.bincount(res["mode"], minlength=9), "failed", int((res["status"] != 0).sum()),
      "kernel ms", np.round(ms / 3, 3), "GCUPS(dp)", round(cells / (ms[1] / 3 * 1e-3) / 1e9, 1), "GCUPS(serial)", round(cells / (ms.sum() / 3 * 1e-3) / 1e9, 1))
om, hs, ccells, bad = mo.yama_batch(batch, variant=1, threads=64)
out = db.out.cpu().numpy(); mism = 0
for i in range(len(batch["K"])):
    w = int(batch["K"][i] + batch["L"][i]); m_, o0 = int(res["om"][i]), int(res["offOut"][i])
    if m_ != om[i] or mo.fnv1a_np(out[o0:o0 + m_ * w], mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8))) != int(hs[i]): mism += 1
print("mismatches", mism)
