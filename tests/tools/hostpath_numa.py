"""mz_yama_batch() from host buffers with the whole process (generator, pool threads, first-touched memory) pinned to
one NUMA node, to none, or to the other: how much of the host path's time is cross-socket traffic.
    python tests/tools/hostpath_numa.py <node|-1> [config]"""
import glob, os, sys, time
node = int(sys.argv[1]) if len(sys.argv) > 1 else -1
if node >= 0:
    cpus = []
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, cpus)
import numpy as np
sys.path.insert(0, '.')
import multiz_amd as mz
from multiz_amd import synth, api
mz.api.init(0)
cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
c = synth.CONFIGS[cfg]
n = c["pairs"]
batch = synth.make_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=0, indel=c.get("indel", 0))
jobs, outs = api.host_jobs(batch)
cells = synth.band_cells(batch)
gpu_nodes = [open(f).read().strip() for f in glob.glob("/sys/class/drm/card*/device/numa_node")]
ts = []
for rep in range(7):
    t = time.perf_counter()
    assert api.yama_batch_records(jobs, outs) == 0
    ts.append(time.perf_counter() - t)
    api.free_outs(outs)
    time.sleep(0.05)
print(f"node {node} (GPU numa_node files: {gpu_nodes}) {cfg}: " + " ".join(f"{1e3*t:.2f}" for t in ts) + f" ms; median {cells/np.median(ts[1:])/1e9:.1f} GCUPS", flush=True)
