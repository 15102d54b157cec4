"""pairs saved by a stress tool (A<j>, B<j>, LB<j>, RB<j> in an .npz) through the device-resident path again: mode, status, final
scores and merged columns against the oracle.
    python tests/tools/replay_npz.py <file.npz> [max pairs]       (MZ_LIB_PATH=<libmzamd.so> picks another build)"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import multiz_amd as mz
from multiz_amd import synth
from oracle import mzoracle as mo
mz.api.init(0)
z = np.load(sys.argv[1])
n = len([k for k in z.files if k.startswith("A")])
if len(sys.argv) > 2: n = min(n, int(sys.argv[2]))
pairs = [(z[f"A{j}"], z[f"B{j}"], z[f"LB{j}"], z[f"RB{j}"]) for j in range(n)]
db = mz.DevBatch(synth.pack_pairs(pairs)); db.run(); res = db.results(); out = db.out.cpu().numpy()
for j, (A, B, LB, RB) in enumerate(pairs):
    w = mo.yama(A, B, LB, RB)
    m_, o0 = int(res["om"][j]), int(res["offOut"][j]); W = A.shape[1] + B.shape[1]
    same = res["status"][j] == 0 and m_ == w.OM and np.array_equal(out[o0:o0 + m_ * W].reshape(m_, W), w.cols)
    wid = (RB.astype(int) - LB + 1)
    viol = max((int(RB[r + 1]) - int(LB[r + 64]) - 62 for r in range(len(LB) - 64)), default=0)
    print(j, "shape", A.shape, B.shape, "mode", int(res["mode"][j]), "status", int(res["status"][j]), "same", bool(same),
          "final3 gpu", res["final3"][j].tolist(), "oracle", list(w.final), "OM", m_, w.OM, "max width", int(wid.max()), "V", viol)
