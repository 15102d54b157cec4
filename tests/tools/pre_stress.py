"""mz_preyama_batch() over one configuration's batch, again and again in one process (one-stage and two-stage merges in turn): a crash or
a call that reports an error ends it.   python tests/tools/pre_stress.py <config> <calls> [pairs]"""
import sys, time, faulthandler
sys.path.insert(0, '.')
faulthandler.enable(all_threads=True)
import numpy as np
import multiz_amd as mz
from multiz_amd import api, synth
mz.api.init(0)
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c4i"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = synth.CONFIGS[cfgname]
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else cfg["pairs"]
pbs = [synth.make_pre_batch(pairs, cfg["K"], cfg["L"], cfg["mlo"], cfg["mhi"], cfg["radius"], events=cfg.get("indel", 0), v=v) for v in (1, 0)]
t0 = time.time()
for i in range(calls):
    pb = pbs[i & 1]
    rc = api.preyama_batch_records(pb["jobs"], pb["outs"])
    api.free_preouts(pb["outs"])
    if (i + 1) % 20 == 0: print(i + 1, "calls", round(time.time() - t0, 1), "s, last rc", rc, flush=True)
print("ok:", calls, "calls")
