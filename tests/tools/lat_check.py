"""k_dp_row_lat (MZ_LAT_MAX) against the oracle on pairs of growing length: which lengths / modes fail.
    python tests/tools/lat_check.py [pairs]"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import multiz_amd as mz
from multiz_amd import synth
from oracle import mzoracle as mo
mz.api.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for lo, hi in ((300, 400), (1000, 1200), (3000, 3300), (8000, 8500), (20000, 21000), (50000, 52000), (95000, 105000)):
    batch = synth.make_batch(n, 2, 2, lo, hi, 30, first_pair=3)
    db = mz.DevBatch(batch); db.run(); res = db.results()
    om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=16)
    out = db.out.cpu().numpy()
    badp = []
    for i in range(n):
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        if m_ != om[i] or mo.fnv1a_np(out[o0:o0 + m_ * 4], mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8))) != int(hs[i]):
            badp.append((i, int(res["mode"][i]), int(batch["M"][i]), int(batch["N"][i]), m_, int(om[i])))
    print(lo, hi, "status ok" if (res["status"] == 0).all() else "STATUS", "bad:", len(badp), badp[:4], flush=True)
