import sys, numpy as np, time
sys.path.insert(0, '.')
import multiz_amd as mz
from multiz_amd import synth
from oracle import mzoracle as mo
mz.api.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
c = synth.CONFIGS["c2"]
batch = synth.make_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=0)
om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=64)
def fnv(a, m):
    return mo.fnv1a_np(a, m) if hasattr(mo, 'fnv1a_np') else None
for row in (int(sys.argv[2]) if len(sys.argv) > 2 else 1,):
    mz.lib().mz_enable_row(row)
    db = mz.DevBatch(batch)
    db.run()
    res = db.results()
    print("row", row, "modes", np.bincount(res["mode"], minlength=9), "failed", int((res["status"] != 0).sum()))
    ms = np.zeros(4)
    for _ in range(5): ms += np.array(db.run(timed=True))
    print("  kernel ms", ms / 5, "GCUPS(dp)", cells / (ms[1] / 5 * 1e-3) / 1e9)
    out = db.out.cpu().numpy()
    bad_i = []
    for i in range(n):
        K, L = int(batch["K"][i]), int(batch["L"][i])
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        if m_ != om[i] or mo.fnv1a_np(out[o0:o0 + m_ * (K + L)], mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8))) != int(hs[i]):
            bad_i.append(i)
    print("  mismatches", len(bad_i), bad_i[:10])
