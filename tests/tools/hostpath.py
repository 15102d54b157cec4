"""mz_yama_batch() from host buffers: wall time and GCUPS of a config's batch, three repetitions after a warm-up.
MZ_TIMING=1|2 in the environment adds the library's JSON lines (per call / per chunk) on stderr.
    python tests/tools/hostpath.py [pairs] [config]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import multiz_amd as mz
from multiz_amd import synth, api
mz.api.init(0)
cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
c = synth.CONFIGS[cfg]
n = int(sys.argv[1]) if len(sys.argv) > 1 and int(sys.argv[1]) > 0 else c["pairs"]
batch = synth.make_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=0, indel=c.get("indel", 0))
jobs, outs = api.host_jobs(batch)
nb = int(batch["offBand"][-1]) + int(batch["M"][-1]) + 1
cells = int((batch["poolRB"][:nb].astype(np.int64) - batch["poolLB"][:nb] + 1).sum())
import os
def vm():
    want = ("numa_hint_faults", "numa_pages_migrated", "pgmigrate_success", "thp_migration_success", "numa_pte_updates")
    d = {k: int(v) for k, v in (l.split() for l in open("/proc/vmstat")) if k in want}
    try:                                                 # the cgroup's CPU quota: periods in which it ran out, time spent throttled
        for l in open("/sys/fs/cgroup/cpu.stat"):
            k, v = l.split()
            if k in ("nr_throttled", "throttled_usec", "nr_periods"): d[k] = int(v)
    except OSError:
        pass
    try:                                                 # this process: context switches it did not ask for
        for l in open("/proc/self/status"):
            if l.startswith("nonvoluntary_ctxt_switches"): d["main_thread_preempted"] = int(l.split()[1])
    except OSError:
        pass
    return d
for rep in range(int(os.environ.get("HOSTPATH_REPS", "4"))):
    v0 = vm()
    t = time.perf_counter()
    rc = api.yama_batch_records(jobs, outs)
    dt = time.perf_counter() - t
    assert rc == 0
    api.free_outs(outs)
    v1 = vm()
    extra = " ".join(f"{k}+{v1[k] - v0[k]}" for k in v0 if v1[k] != v0[k]) if os.environ.get("HOSTPATH_VM") else ""
    print(f"mz_yama_batch({n} {cfg} pairs, host buffers in, merged columns out): {dt*1e3:.2f} ms -> {cells/dt/1e9:.1f} GCUPS {extra}", flush=True)
    time.sleep(float(os.environ.get("HOSTPATH_SLEEP", "0")))
