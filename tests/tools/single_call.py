# latency of the drop-in path: one block pair per mz_yama_batch() call (what the unmodified multiz driver does)
import sys, time, ctypes as C, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import inputs
import multiz_amd as mz
from multiz_amd import api
from oracle import mzoracle as mo
mz.api.init(0)
rng = np.random.default_rng(3)
A, B, LB, RB = inputs.make_pair(rng, 3, 3, 200, 210, 30, "diag", mo.smooth)
job = (api.Job * 1)(); out = (api.Out * 1)()
job[0].K, job[0].L, job[0].M, job[0].N = 3, 3, 200, 210
job[0].A, job[0].B, job[0].LB, job[0].RB = A.ctypes.data, B.ctypes.data, LB.ctypes.data, RB.ctypes.data
for rep in range(3):
    t = time.perf_counter()
    for _ in range(2000):
        mz.lib().mz_yama_batch(1, job, out)
        mz.lib().free_cols(out[0].cols)   # (n == 1: the block itself)
    dt = time.perf_counter() - t
    print(f"one 200x210 pair (3+3 rows) per call: {dt / 2000 * 1e6:.1f} us per call")
