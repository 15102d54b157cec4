"""Does the ORDER in which k_dp_lag's waves take their pairs matter?  The c2i batch as generated, sorted by rows
descending (longest first) and ascending; the DP time of each (db.run(timed=True): plan, dp, walk, emit in ms).

    python tests/tools/lag_order.py [config] [reps]
"""
import sys, numpy as np
sys.path.insert(0, '.')
import multiz_amd as mz
from multiz_amd import synth

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2i"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mz.api.init(0)
c = synth.CONFIGS[cfgname]
batch = synth.make_batch(c["pairs"], c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], indel=c.get("indel", 0))
offs = batch["offBand"]
w = (batch["poolRB"].astype(np.int64) - batch["poolLB"] + 1)
cs = np.concatenate([[0], np.cumsum(w)])
cells = cs[offs + batch["M"] + 1] - cs[offs]
rows = batch["M"].astype(np.int64)
for name, idx in (("as generated", np.arange(len(cells))), ("longest first", np.argsort(-rows, kind="stable")),
                  ("most cells first", np.argsort(-cells, kind="stable")), ("shortest first", np.argsort(rows, kind="stable"))):
    sub = synth.subset(batch, idx)
    db = mz.DevBatch(sub); db.run()
    ms = np.array([db.run(timed=True) for _ in range(reps)])
    print(f"{cfgname} {name:15s} dp ms median {np.median(ms[:,1]):.3f} min {ms[:,1].min():.3f}   all phases {np.round(np.median(ms, axis=0), 3)}")
    del db
