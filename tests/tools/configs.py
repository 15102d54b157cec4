import sys, numpy as np
sys.path.insert(0, '.')
import multiz_amd as mz
from multiz_amd import synth
from oracle import mzoracle as mo
mz.api.init(0)
cfgname = sys.argv[1]; n = int(sys.argv[2])
c = synth.CONFIGS[cfgname]
batch = synth.make_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=0)
for row in (1, 0):
    mz.lib().mz_enable_row(row)
    db = mz.DevBatch(batch)
    db.run()
    res = db.results()
    cells = int(res["cells"].sum())
    ms = np.zeros(4)
    for _ in range(3): ms += np.array(db.run(timed=True))
    print(cfgname, "row", row, "modes", np.bincount(res["mode"], minlength=9), "failed", int((res["status"] != 0).sum()),
          "kernel ms", np.round(ms / 3, 3), "GCUPS(dp)", round(cells / (ms[1] / 3 * 1e-3) / 1e9, 1), "GCUPS(serial)", round(cells / (ms.sum() / 3 * 1e-3) / 1e9, 1))
    if row == 1:
        keep = res
        out1 = db.out.cpu().numpy()
    else:
        out0 = db.out.cpu().numpy()
        same = all(np.array_equal(out1[int(keep["offOut"][i]):int(keep["offOut"][i]) + int(keep["om"][i]) * (c["K"] + c["L"])],
                                  out0[int(res["offOut"][i]):int(res["offOut"][i]) + int(res["om"][i]) * (c["K"] + c["L"])]) for i in range(n))
        print("  row-parallel output == wavefront output:", same)
