"""Packing throughput of mz_yama_batch()'s host side on this machine, no GPU in the loop (mz_host_pack_probe).
    python tests/tools/hostprobe.py [config]      MZ_HOST_THREADS=<n> sets the pool size"""
import ctypes as C, os, sys, numpy as np
os.environ["MZ_NO_TORCH"] = "1"
sys.path.insert(0, ".")
from multiz_amd import synth, api
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
c = synth.CONFIGS[cfg]
for n in (4167, c["pairs"]):
    batch = synth.make_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"])
    jobs, outs = api.host_jobs(batch)
    f = api.lib().mz_host_pack_probe
    f.restype = C.c_double
    f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
    res = {w: f(n, jobs.ctypes.data, w, 20 if n < 10000 else 5) for w in (1, 2, 3)}
    inb = int((batch["K"].astype(np.int64) * batch["M"] + batch["L"].astype(np.int64) * batch["N"] + 8 * (batch["M"].astype(np.int64) + 1)).sum())
    print(f"{cfg} {n} pairs, MZ_HOST_THREADS={os.environ.get('MZ_HOST_THREADS', 'default')}: all {1e3*res[1]:.3f} ms ({inb/res[1]/1e9:.0f} GB/s read), "
          f"classes {1e3*res[2]:.3f} ms, band {1e3*res[3]:.3f} ms; per pair {1e9*res[1]/n:.0f} ns wall", flush=True)
