import sys, time, ctypes as C, numpy as np
sys.path.insert(0, '.')
import multiz_amd as mz
from multiz_amd import synth, api
mz.api.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
c = synth.CONFIGS[cfg]
batch = synth.make_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=0)
jobs = (api.Job * n)(); outs = (api.Out * n)()
for i in range(n):
    jobs[i].K, jobs[i].L, jobs[i].M, jobs[i].N = int(batch["K"][i]), int(batch["L"][i]), int(batch["M"][i]), int(batch["N"][i])
    jobs[i].A = batch["poolA"].ctypes.data + int(batch["offA"][i])
    jobs[i].B = batch["poolB"].ctypes.data + int(batch["offB"][i])
    jobs[i].LB = batch["poolLB"].ctypes.data + 4 * int(batch["offBand"][i])
    jobs[i].RB = batch["poolRB"].ctypes.data + 4 * int(batch["offBand"][i])
cells = int((batch["poolRB"].astype(np.int64) - batch["poolLB"] + 1)[: int(batch["offBand"][-1]) + int(batch["M"][-1]) + 1].sum())
for rep in range(3):
    t = time.perf_counter()
    rc = mz.lib().mz_yama_batch(n, jobs, outs)
    dt = time.perf_counter() - t
    assert rc == 0
    for i in range(n): mz.lib().free_cols(outs[i].cols)
    print(f"mz_yama_batch({n} {cfg} pairs, host buffers in, malloc'ed columns out): {dt*1e3:.1f} ms -> {cells/dt/1e9:.1f} GCUPS")
