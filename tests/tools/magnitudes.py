import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import inputs
import multiz_amd as mz
from multiz_amd import synth
from oracle import mzoracle as mo
mz.api.init(0)
def checker(K, L, M, N, R, pat):
    # columns whose dash pattern flips from one column to the next: every row pair opens a gap at every step
    r = np.arange(M)[:, None]; i = np.arange(K)[None, :]
    A = np.where(((r + i) % 2 == 0) if pat == 0 else ((r // 2 + i) % 2 == 0), ord('A'), ord('-')).astype(np.uint8)
    c = np.arange(N)[:, None]; j = np.arange(L)[None, :]
    B = np.where(((c + j) % 2 == 1), ord('C'), ord('-')).astype(np.uint8)
    if K == 1: A[:] = ord('A')
    if L == 1: B[:] = ord('C')
    for X in (A, B):
        alld = (X == 45).all(axis=1)
        X[alld, 0] = ord('G')
    LB, RB = inputs.diag_band(M, N)
    LB, RB = mo.smooth(LB, RB, M, N, R)
    return A, B, LB, RB
pairs = []
for (K, L, M, N, R) in ((8, 8, 740, 740, 30), (8, 8, 700, 760, 30), (16, 16, 700, 700, 30), (20, 20, 900, 860, 30), (30, 30, 420, 400, 30),
                        (4, 4, 2900, 2900, 30), (2, 2, 11000, 11000, 30), (12, 10, 1500, 1400, 25)):
    for pat in (0, 1):
        pairs.append(checker(K, L, M, N, R, pat))
batch = synth.pack_pairs(pairs)
for row in (1, 0):
    mz.lib().mz_enable_row(row)
    db = mz.DevBatch(batch); db.run(); res = db.results()
    out = db.out.cpu().numpy()
    bad = 0
    for i, (A, B, LB, RB) in enumerate(pairs):
        want = mo.yama(A, B, LB, RB)
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        ok = res["status"][i] == 0 and m_ == want.OM and np.array_equal(out[o0:o0 + m_ * (A.shape[1] + B.shape[1])].reshape(m_, -1), want.cols)
        bad += not ok
        if row == 1:
            print(i, A.shape, B.shape, "mode", int(res["mode"][i]), "final", res["final3"][i], "oracle", want.final, "OK" if ok else "BAD")
    print("row", row, "bad", bad, "modes", np.bincount(res["mode"], minlength=9))
