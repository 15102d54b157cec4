"""What the GPU box's host CPUs give the CPU baseline: cgroup quota, topology, and the compiled reference's yama()
(oracle/_ref/libref.so through oracle/ref_batch.c; the oracle's faithful port when absent) at 1..all threads on
C2 pairs -- thread-scaling curve, process CPU time against wall time (a quota or descheduling shows as
cpu_s / (wall * threads) << 1), and a pure spin loop (no memory) at the same thread counts.
Never touches the GPU.   python tests/tools/cpu_scaling.py [config] [seconds per point]"""
import json
import os
import resource
import sys
import time

os.environ["MZ_NO_TORCH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def box_info():
    info = {"nproc_affinity": len(os.sched_getaffinity(0)), "cpu_count": os.cpu_count()}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
              "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpu.stat"):
        v = read(p)
        if v is not None:
            info[p] = v
    model, mhz = None, []
    for line in (read("/proc/cpuinfo") or "").splitlines():
        if line.startswith("model name") and model is None:
            model = line.split(":", 1)[1].strip()
        if line.startswith("cpu MHz"):
            mhz.append(float(line.split(":", 1)[1]))
    info["model"] = model
    if mhz:
        info["mhz_min_max"] = [min(mhz), max(mhz)]
    pk, cores = set(), set()
    for c in sorted(os.sched_getaffinity(0)):
        b = "/sys/devices/system/cpu/cpu%d/topology/" % c
        p_, c_ = read(b + "physical_package_id"), read(b + "core_id")
        pk.add(p_)
        cores.add((p_, c_))
    info["packages"] = len(pk)
    info["physical_cores"] = len(cores)
    info["hypervisor"] = "hypervisor" in (read("/proc/cpuinfo") or "")
    return info


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "c2"
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    print(json.dumps(box_info()))
    import ctypes as C
    import numpy as np
    from multiz_amd import synth
    from oracle import mzoracle as mo
    cfg = synth.CONFIGS[config]
    ncpu = len(os.sched_getaffinity(0))
    batch = synth.make_batch(4096, cfg["K"], cfg["L"], cfg["mlo"], cfg["mhi"], cfg["radius"], indel=cfg.get("indel", 0))
    use_ref = mo.have_reference()
    run = (lambda bt, th: mo.ref_batch(bt, threads=th)) if use_ref else (lambda bt, th: mo.yama_batch(bt, variant=0, threads=th))
    one = synth.subset(batch, np.arange(4))
    t = time.perf_counter(); run(one, 1); per_pair = (time.perf_counter() - t) / 4
    spin = mo.lib().mzo_spin if hasattr(mo.lib(), "mzo_spin") else None
    if spin is not None:
        spin.restype = C.c_double
        spin.argtypes = [C.c_int, C.c_int64]
    counts = sorted({1, 2, 4, 8, 16, 32, 64, 128, ncpu} & set(range(1, ncpu + 1)))
    base = None
    for th in counts:
        npairs = int(min(4096, max(th, seconds * th / per_pair)))
        sub = synth.subset(batch, np.arange(npairs))
        run(synth.subset(batch, np.arange(min(npairs, 2 * th))), th)          # thread pool warm
        r0 = resource.getrusage(resource.RUSAGE_SELF)
        t = time.perf_counter()
        _, _, cells, _ = run(sub, th)
        wall = time.perf_counter() - t
        r1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu_s = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
        g = cells / wall / 1e9
        base = base or g
        line = {"threads": th, "pairs": npairs, "gcups": round(g, 5), "speedup": round(g / base, 2), "efficiency": round(g / base / th, 3),
                "cpu_s_over_wall_threads": round(cpu_s / (wall * th), 3), "sys_frac": round((r1.ru_stime - r0.ru_stime) / max(cpu_s, 1e-9), 3)}
        if spin is not None:
            line["spin_s"] = round(spin(th, 400_000_000), 3)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
