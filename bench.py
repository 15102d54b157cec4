#!/usr/bin/env python3
"""bench.py -- GCUPS (DP cell updates / s) of the yama block-pair merge on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5] [--scatter]

A "step" is one pass of the whole hot path -- validity/plan, banded DP with traceback, traceback walk,
merged columns (reference mz_yama.c:58-313) -- over one batch of synthetic block pairs.  Workload at N=1:
BASELINE.json configs[1] ("50k synthetic block pairs, 2+2 rows, ~1k x 1k cols, banded yama DP on 1 MI355X");
--config picks another BASELINE configuration (c3 deep sum-of-pairs, c4 one GPU's share of the 30-leaf guide-tree
workload, c5 long blocks).  With N > 1 every rank runs its own shard of the same generator (weak scaling, no
data-path collective: block pairs are independent; one all-reduce of three scalars closes the batch).  --scatter
instead builds the WHOLE list on rank 0 and deals it out through the library's exchange (include/mz_shard.h: RCCL
grouped send / recv, chunks of the list packed, moved, aligned and assembled side by side) before the timed steps --
BASELINE configs[3]'s "sharded over 8 x MI355X via RCCL" as one command; the exchange times are reported beside the
value.  `--gpus N` without a torch.distributed environment starts the N ranks itself.

One JSON line on stdout (rank 0).  SURVEY.md section 8(d) defines the metric's wall time as "H2D of packed jobs +
kernels + D2H of packed outputs (report kernel-only as a second column)", so -- since round 6 --

    value / ms_per_step   K calls of mz_yama_batch(): the batch from HOST buffers to malloc()ed merged columns -- packing,
                          PCIe both ways and the host-side assembly included; what the reference's drivers get.  Every call
                          is timed on its own between barrier + synchronize (max over ranks); value = cells x K / the sum
                          of the K times (`value_host`: the same number; `host_median`: what rounds 3-5 printed).  The
                          calls are 50 ms apart, outside the timed brackets: the GPU boxes cap a job at 16 CPUs' worth of
                          time per 100 ms and calls issued back to back run into that quota (`back_to_back`: that rate
                          with the cgroup's throttling counters -- faster than the calls apart where the quota holds).
    value_resident        the second column: the batch already in HBM, the pipelined device-resident form
                          (mz_dev_run_async), `resident_steps` steps -- what rounds 1-5 printed as `value`.
    single_batch_gcups    one resident batch, its phases one after the other from HIP events; `roofline` is the DP launch
                          of that form.
    value_pre / _v0       the same work as block TEXT through mz_preyama_batch() (what mz_multiz / mz_roast use).

Cells are band cells, counted exactly as the reference counts tback_size (mz_yama.c:60-66).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The pipelined form keeps up to nine HIP streams busy (DPs of consecutive small batches side by side, plan, walk + emit);
# the runtime maps streams onto 4 hardware queues by default and streams sharing a queue serialise.  Must be in the
# environment before the HIP runtime starts (torch starts it here, before the library could).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# VALU issue (tests/tools/ub/ops.hip, raw output in profiles/r2_ub_ops.txt): at two or more waves per SIMD v_add_u32 / v_sub_u32 /
# v_and_b32 occupy a SIMD for 2.3-2.5 shader cycles per wave-instruction, every other integer operation of this path (max, max3,
# and_or, alignbit, dot2 / dot4, DPP forms, perm, bfe, mad24, compares) for 4.0-4.2.  A launch is priced by the static mix of its
# kernel's loops (profiles/r*_isa_hist.json, tests/tools/isa_hist.py: from the compiler's own assembly); without that record every
# instruction is priced at VALU_CYCLES_REST -- an overestimate of the issue time, never an underestimate.  1024 SIMDs.
VALU_CYCLES_FAST, VALU_CYCLES_REST = 2.35, 4.1
SIMDS = 1024


def isa_mix(kernel):
    """cycles per VALU wave-instruction of `kernel` from the newest profiles/r*_isa_hist.json; (cycles, source, stale)"""
    import glob
    import re

    def round_of(f):
        m = re.match(r"r(\d+)", os.path.basename(f))
        return (int(m.group(1)) if m else -1, os.path.basename(f))
    cur, found = sources_hash(), None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_hist.json")), key=round_of, reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        k = d.get("kernels", {}).get(kernel, {}).get("in_loops")
        if not k or not k.get("valu"):
            continue
        rec = (float(k["cycles_per_inst"]), os.path.relpath(f, ROOT), d.get("sources_hash") != cur, {"valu": k["valu"], "fast": k["fast"]})
        if not rec[2]:
            return rec
        found = found or rec
    return found or (VALU_CYCLES_REST, None, False, None)


def sources_hash():
    """what the PMC records are keyed by: the device sources (same function as tests/tools/pmc_collect.py)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "multiz_amd", "csrc")
    for f in [os.path.join(base, "mz_device.hip")] + sorted(glob.glob(os.path.join(base, "kernels", "*.inc"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_record(config, pairs, kernel):
    """PMC figures of one launch of `kernel` on this config's batch from the newest profiles/r*_pmc.json (collected by
    tests/tools/pmc_collect.py: separate rocprofv3 --pmc passes, FETCH_SIZE doubled as the guide prescribes for gfx950).
    Measured, never estimated; "stale": the device sources have changed since they were collected."""
    import glob
    import re

    def round_of(f):                                   # r10 after r9 (a plain sort puts "r10" before "r3")
        m = re.match(r"r(\d+)", os.path.basename(f))
        return (int(m.group(1)) if m else -1, os.path.basename(f))
    cur, found = sources_hash(), None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc*.json")), key=round_of, reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        r = d.get("records", {}).get(f"{config}:{pairs}", {}).get(kernel)
        if not r or "SQ_INSTS_VALU" not in r or r.get("incomplete"):
            continue
        rec = dict(r, source=os.path.relpath(f, ROOT), stale=d.get("sources_hash") != cur)
        if not rec["stale"]:
            return rec                                 # a record of the current device sources wins over a newer stale one
        found = found or rec
    return found


def spawn_ranks(args):
    """--gpus N given, no torch.distributed environment: start the N ranks as child processes (before this process
    touches the GPU) and exit with their status"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


# ------------------------------------------------------------------------------------------ CPU leg (child process)

def one_socket_cores():
    """one logical CPU per physical core of the socket CPU 0 sits on (sysfs topology); model name"""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        allowed = sorted(os.sched_getaffinity(0))
        pkg0 = open("/sys/devices/system/cpu/cpu%d/topology/physical_package_id" % allowed[0]).read().strip()
        seen, cpus = set(), []
        for c in allowed:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            if open(base + "physical_package_id").read().strip() != pkg0:
                continue
            core = open(base + "core_id").read().strip()
            if core not in seen:
                seen.add(core)
                cpus.append(c)
        return cpus, model
    except OSError:
        return sorted(os.sched_getaffinity(0)), model


def cpu_throttle():
    """(nr_throttled, throttled_usec) of this container's cgroup, or None: the CPU quota at work -- when the job has used its
    quota of a 100 ms period every thread of it is parked until the next one, the threads that feed the GPU included"""
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except (OSError, ValueError):
        return None


def cpu_budget():
    """CPUs' worth of time the container may use: the affinity mask capped by the cgroup quota (cpu.max); the quota text"""
    n = len(os.sched_getaffinity(0))
    text = None
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            text = open(path).read().strip()
            q, per = text.split()
            if q != "max":
                n = min(n, max(1, -(-int(q) // int(per))))
        except (OSError, ValueError):
            pass
    return n, text


def cpu_leg(spec_path):
    """Child process, never touches the GPU: the reference's yama() (oracle/_ref/libref.so; the oracle's faithful
    restatement when that is absent) on a seeded sample of the same batch, one pair per OpenMP thread.

    Thread counts 1, 2, 4, ... up to the physical cores of ONE socket are timed on ~2 s samples each (the scaling curve:
    the GPU boxes of this project cap a job at 16 CPUs' worth of time -- cpu.max "1600000 100000" -- so 64 threads deliver
    what 16 do; round 2's "0.42 GCUPS on 64 cores" was that quota, not the reference).  The headline sample then runs at the
    best count.  `socket_linear` = that rate x (cores of the socket / threads), an EXTRAPOLATION justified by the measured
    efficiency up to the quota and stated as such.  Writes om / hash per sampled pair for the parent's parity gate."""
    spec = json.load(open(spec_path))
    cpus, model = one_socket_cores()
    os.sched_setaffinity(0, cpus)                      # before any OpenMP runtime starts: its threads inherit the mask
    os.environ["OMP_PROC_BIND"] = "spread"
    os.environ["OMP_PLACES"] = "cores"
    import numpy as np
    from multiz_amd import synth
    from oracle import mzoracle as mo
    cfg, pairs, cores = synth.CONFIGS[spec["config"]], spec["pairs"], len(cpus)
    budget, quota_text = cpu_budget()
    batch = synth.make_batch(pairs, cfg["K"], cfg["L"], cfg["mlo"], cfg["mhi"], cfg["radius"], first_pair=spec["first_pair"],
                             indel=cfg.get("indel", 0))
    use_ref = mo.have_reference()
    run_cpu = (lambda bt, th: mo.ref_batch(bt, threads=th)) if use_ref else (lambda bt, th: mo.yama_batch(bt, variant=0, threads=th))
    rng = np.random.default_rng(12345)
    order = rng.permutation(pairs)
    one = synth.subset(batch, np.sort(order[: min(pairs, 4)]))
    run_cpu(one, 1)                                    # warm-up: page faults, the library itself
    t = time.perf_counter()
    run_cpu(one, 1)
    per_pair = (time.perf_counter() - t) / len(one["K"])
    # ---- the scaling curve (a third of the budget)
    counts = sorted({c for c in (1, 2, 4, 8, 16, 32, 64, 128) if c < cores} | {cores})
    per_point = spec["seconds"] / 3.0 / len(counts)
    curve, base = [], None
    for th in counts:
        eff_th = min(th, budget)
        k = int(max(th, min(pairs, per_point * eff_th / max(per_pair, 1e-7))))
        sub = synth.subset(batch, np.sort(order[:k]))
        run_cpu(synth.subset(batch, np.sort(order[: min(k, 2 * th)])), th)      # thread pool up
        t = time.perf_counter()
        _, _, c_, _ = run_cpu(sub, th)
        g = c_ / (time.perf_counter() - t) / 1e9
        base = base or g
        curve.append({"threads": th, "gcups": round(g, 5), "efficiency": round(g / base / th, 3)})
    # the FEWEST threads within 3 % of the best rate: past the quota more threads give the same rate (or less: the quota
    # throttles all of them), and socket_linear = rate x cores / threads must not be halved by a tie
    top = max(r["gcups"] for r in curve)
    best = min((r for r in curve if r["gcups"] >= 0.97 * top), key=lambda r: r["threads"])
    # ---- the headline sample at the best thread count (its hashes are the parity gate)
    nsample = int(max(min(cores, pairs), min(pairs, spec["seconds"] * 2.0 / 3.0 * best["gcups"] / max(curve[0]["gcups"], 1e-9) / max(per_pair, 1e-7))))
    idx = np.sort(order[:nsample])
    sample = synth.subset(batch, idx)
    t = time.perf_counter()
    om, hs, ccells, bad = run_cpu(sample, best["threads"])
    cpu_s = time.perf_counter() - t
    np.savez(spec["out"], idx=idx, om=om, hs=hs)
    value = ccells / cpu_s / 1e9
    what = ("reference yama() (oracle/_ref/libref.so, gcc -O2 -fcommon)" if use_ref
            else "oracle faithful O(K*L)/cell restatement (gcc -O2)")
    print(json.dumps({"value": round(value, 5), "unit": "GCUPS", "cores": best["threads"], "model": model,
                      "kind": "reference" if use_ref else "port", "bad": int(bad), "one_thread": curve[0]["gcups"],
                      "socket_cores": cores, "cpu_max": quota_text, "effective_cpus": budget, "scaling": curve,
                      "socket_linear": round(value * cores / best["threads"], 4),
                      "sample": f"{nsample} of the {pairs} pairs (seeded), {ccells} band cells in {cpu_s:.1f} s on {best['threads']} threads; {what}, "
                                f"OpenMP one pair per thread, pinned to the {cores} physical cores of one socket; the container's CPU quota "
                                f"is {budget} CPUs (cpu.max = {quota_text}): socket_linear extrapolates the measured rate to all {cores} cores"}))


# ------------------------------------------------------------------------------------------ one process, N GPUs

def ngpu_mode(args):
    """`--mode ngpu --gpus N`: ONE process; the library opens N contexts (mz_init_multi) and mz_yama_batch() deals a host
    list of N x pairs out over them -- contiguous ranges of about equal weight, one host thread and one PCIe link per GPU,
    no exchange between GPUs (the path mz_multiz / mz_roast take on a node with MZ_NGPU=N).  The rate includes packing,
    both PCIe directions and the assembly of the merged columns on the host.  MZ_ALLOW_DUP_DEVICES=1 (tests): the N
    contexts on GPU 0."""
    import numpy as np
    from multiz_amd import api, synth
    cfg = dict(synth.CONFIGS[args.config])
    pairs = (args.pairs or cfg["pairs"]) * args.gpus
    dup = os.environ.get("MZ_ALLOW_DUP_DEVICES") == "1"
    api.init_multi(args.gpus, [0] * args.gpus if dup else None)
    ngpu = api.lib().mz_device_count()
    idents = [api.device_identity(i) for i in range(ngpu)]
    batch = synth.make_batch(pairs, cfg["K"], cfg["L"], cfg["mlo"], cfg["mhi"], cfg["radius"], indel=cfg.get("indel", 0))
    cells = synth.band_cells(batch)
    jobs, outs = api.host_jobs(batch)
    for _ in range(max(1, args.warmup)):
        api.yama_batch_records(jobs, outs)
        om = outs["OM"].copy(); ok = bool((outs["status"] == 0).all())
        api.free_outs(outs)
    times = []
    steps = min(args.steps, 10)
    for _ in range(steps):
        t = time.perf_counter()
        api.yama_batch_records(jobs, outs)
        times.append(time.perf_counter() - t)
        assert np.array_equal(outs["OM"], om)
        api.free_outs(outs)
    if not ok:
        raise SystemExit("block pairs failed on the device -- number void")
    link = api.link_bytes(jobs)
    t_med = float(np.median(times))
    print(json.dumps({
        "metric": "GCUPS (DP cell updates/s) on yama block-pair merge", "mode": "ngpu",
        "value": round(cells / t_med / 1e9, 2), "unit": "GCUPS", "n_gpus": ngpu, "steps": steps, "warmup": max(1, args.warmup),
        "ms_per_step": round(1e3 * t_med, 3), "ms_all": [round(1e3 * x, 2) for x in times], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": synth.describe(args.config, pairs // args.gpus), "pairs_total": pairs, "band_cells_total": cells,
                   "parallelism": f"one process, mz_init_multi({ngpu}): a host list dealt over {ngpu} GPU contexts, one host thread and "
                                  f"one PCIe link each (host buffers in, merged columns out: transfers included)"},
        "devices": idents, "distinct_devices": len(set(idents)),
        "link_bytes_per_pair": {"up": round(link[0] / pairs, 1), "down": round(link[1] / pairs, 1)}}))


# ------------------------------------------------------------------------------------------ block text in, block rows out

def pre_check(pb, outs, idx, radius):
    """parity gate of the text path: the sampled merges through the compiled reference's pre_yama() (oracle/_ref/libref.so;
    the oracle's Python restatement when that is absent) -- rows, base counts and score of every block.  Checker code."""
    import numpy as np
    from multiz_amd import api, synth
    from oracle import mzoracle as mo
    ref = mo.Reference() if mo.have_reference() else None
    bad = 0
    for i in idx:
        i = int(i)
        r1, r2 = synth.pre_rows_of(pb, i)
        v = int(pb["jobs"]["v"][i])

        def blk(rows, names):
            return mo.Block(rows=[mo.Row(src=nm, start=0, size=sum(c != 0x2D for c in t), strand="+", srcSize=1 << 24, text=t.decode("ascii"))
                                  for nm, t in zip(names, rows)])
        a1 = blk(r1, ["ref.c"] + [f"a{k}.c" for k in range(1, len(r1))])
        a2 = blk(r2, ["ref.c"] + [f"b{k}.c" for k in range(1, len(r2))])
        end = a1.rows[0].size - 1
        want = ref.pre_yama(a1, a2, 0, end, radius, v) if ref else mo.pre_yama(a1, a2, 0, end, radius, v)[0]
        rows, size = api.preout_rows(outs, i, len(r1) + len(r2) - 1)
        if want is None or rows is None:
            bad += not (want is None and rows is None)
            continue
        got = [(t.decode("ascii"), int(sz)) for t, sz in zip(rows, size) if sz > 0]
        if got != [(r.text, r.size) for r in want.rows] or float(outs["score"][i]) != want.score:
            bad += 1
    return bad, ("reference pre_yama() (oracle/_ref/libref.so)" if ref else "oracle restatement of pre_yama()")


def pre_column(config, pairs, check):
    """`value_pre`: the batch as block TEXT through mz_preyama_batch() -- class nibbles up, rmColDash + band + smooth + yama()
    (both of them for v = 0) + base counts + mafScoreRange on the GPU, a record and column bits back, the merged block's rows put
    together on the host -- one-stage (v = 1) and two-stage (v = 0) merges; median of five calls each.  The text shapes are the
    yama-level configuration's (synth.make_pre_batch: K rows over L rows below the shared reference row, R reference bases in the
    config's column range; the indel configurations' runs of unshared columns, none otherwise: the band is the diagonal's)."""
    import numpy as np
    from multiz_amd import api, synth
    cfg = synth.CONFIGS[config]
    out = {}
    for v, key in ((1, "v1"), (0, "v0")):
        pb = synth.make_pre_batch(pairs, cfg["K"], cfg["L"], cfg["mlo"], cfg["mhi"], cfg["radius"], events=cfg.get("indel", 0), v=v)
        jobs, outs = pb["jobs"], pb["outs"]
        for _ in range(2):                                       # warm-up: staging buffers and result blocks grow to size
            rc = api.preyama_batch_records(jobs, outs)
            api.free_preouts(outs)
        time.sleep(0.3)
        ts = []
        thr0 = cpu_throttle()
        for _ in range(9):
            t = time.perf_counter()
            rc = api.preyama_batch_records(jobs, outs)
            ts.append(time.perf_counter() - t)
            api.free_preouts(outs)
            time.sleep(0.05)
        thr1 = cpu_throttle()
        up, down, cells = api.pre_link_bytes()
        t_med = float(np.median(ts))
        d = {"gcups": round(cells / t_med / 1e9, 2), "ms_per_batch": round(1e3 * t_med, 2), "ms_all": [round(1e3 * x, 2) for x in ts],
             "cgroup_throttled_during_the_calls": ({"periods": thr1[0] - thr0[0], "usec": thr1[1] - thr0[1]} if thr0 and thr1 else None),
             "band_cells": cells, "merges": int(pairs), "without_block": int(rc), "two_stage_merges": int((jobs["v"] == 0).sum()),
             "link_bytes_per_merge": {"up": round(up / pairs, 1), "down": round(down / pairs, 1)},
             "text_bytes_per_merge": round(float((pb["K"].astype(np.int64) * pb["Ma"] + pb["L1"].astype(np.int64) * pb["Na"]).mean()), 1)}
        if check:
            api.preyama_batch_records(jobs, outs)                # (untimed call: the rows the checker reads)
            idx = np.random.default_rng(777).permutation(pairs)[: min(pairs, check)]
            bad, what = pre_check(pb, outs, idx, cfg["radius"])
            api.free_preouts(outs)
            if bad:
                raise SystemExit(f"PARITY FAILURE (text path, v = {v}): {bad} of {len(idx)} sampled merges differ from the {what} -- number void")
            d["parity"] = f"ok: {len(idx)} sampled merges (rows, base counts, score) identical to the {what}"
        out[key] = d
        del pb
    return out


def pre_mode(args):
    """`--mode pre`: only the text path, `steps` calls of mz_preyama_batch() over one batch (what the rocprofv3 passes behind
    profiles/r4_*pre* run: k_pre / k_mid / k_fin beside the DP kernels)"""
    import numpy as np
    from multiz_amd import api, synth
    api.init(0)
    cfg = synth.CONFIGS[args.config]
    pairs = args.pairs or cfg["pairs"]
    pb = synth.make_pre_batch(pairs, cfg["K"], cfg["L"], cfg["mlo"], cfg["mhi"], cfg["radius"], events=cfg.get("indel", 0), v=args.pre_v)
    ts = []
    for i in range(args.warmup + args.steps):
        t = time.perf_counter()
        api.preyama_batch_records(pb["jobs"], pb["outs"])
        ts.append(time.perf_counter() - t)
        api.free_preouts(pb["outs"])
    up, down, cells = api.pre_link_bytes()
    t_med = float(np.median(ts[args.warmup:]))
    print(json.dumps({"metric": "GCUPS (DP cell updates/s) on pre_yama merges, block text in, block rows out", "mode": "pre",
                      "value": round(cells / t_med / 1e9, 2), "unit": "GCUPS", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": round(1e3 * t_med, 3), "merges": pairs, "two_stage_merges": int((pb["jobs"]["v"] == 0).sum()),
                      "band_cells": cells, "link_bytes_per_merge": {"up": round(up / pairs, 1), "down": round(down / pairs, 1)},
                      "config": {"workload": synth.describe(args.config, pairs) + " -- as block text"}}))


# ------------------------------------------------------------------------------------------ main

def algorithmic_bytes(batch, om):
    """K*M + L*N (columns in) + 8*(M+1) (LB,RB) + cells (1 B traceback per
    cell) + (M+N) (traceback read, upper bound) + (K+L)*OM (merged block out), summed over pairs."""
    import numpy as np
    K, L, M, N = (batch[k].astype(np.int64) for k in ("K", "L", "M", "N"))
    n_band = int(batch["offBand"][-1]) + int(M[-1]) + 1
    cells = int((batch["poolRB"][:n_band].astype(np.int64) - batch["poolLB"][:n_band] + 1).sum())
    total = int((K * M + L * N + 8 * (M + 1) + (M + N) + (K + L) * om.astype(np.int64)).sum()) + cells
    return total, cells


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)    # timed calls of mz_yama_batch() (the headline: SURVEY 8d's transfer-inclusive wall time)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--resident-steps", type=int, default=0,     # (a c2 step is ~5.1 ms; the pipeline's fill and drain cost ~2.5 ms per run)
                    help="timed steps of the device-resident pipeline (value_resident); default 100, with --no-host: --steps")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5", "c2i", "c4i", "c2w", "c2s", "c2g"])
    ap.add_argument("--pairs", type=int, default=0, help="override pairs per GPU (default: the config's)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="budget of the CPU-baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-host", action="store_true", help="skip the host-buffer (PCIe-inclusive) columns")
    ap.add_argument("--no-pre", action="store_true", help="skip the block-text column (value_pre)")
    ap.add_argument("--host-gap-ms", type=float, default=50.0, help="pause between the timed host-path calls (outside the timed brackets; 0: back to back)")
    ap.add_argument("--pre-check", type=int, default=300, help="merges per variant the text path's parity gate samples (0: none)")
    ap.add_argument("--scatter", action="store_true", help="N > 1: rank 0 builds the whole list and deals it out (multiz_amd.shard)")
    ap.add_argument("--pre-v", type=int, default=2, help="--mode pre: 1 one-stage merges, 0 two-stage, 2 alternating")
    ap.add_argument("--mode", default="ranks", choices=["ranks", "ngpu", "pre"],
                    help="ranks: one process per GPU (torch.distributed, the scaling measurement); ngpu: ONE process, the C path "
                         "mz_init_multi(N) + mz_yama_batch() over a host list of N x pairs -- what mz_multiz / mz_roast use on a node")
    ap.add_argument("--cpu-leg", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_leg:
        return cpu_leg(args.cpu_leg)
    if args.mode == "ngpu":
        return ngpu_mode(args)
    if args.mode == "pre":
        return pre_mode(args)

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        spawn_ranks(args)
    world = int(env_world or "1")
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus} (or without a distributed environment, and bench.py starts the ranks)")

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    # ---- CPU baseline first (rank 0, N=1 only), in a child process that never touches the GPU, pinned to one
    # socket's physical cores, before this process initialises the device.  Preferred: the REAL reference's yama()
    # (oracle/_ref/libref.so, built from /root/reference by oracle/Makefile, -O2) driven one pair per thread; otherwise
    # the oracle's faithful O(K*L)/cell restatement.  Checker/baseline only; its per-pair hashes feed the parity gate.
    cpu = cpu_arrays = None
    if rank == 0 and world == 1 and not args.no_cpu:
        import numpy as np
        cfg0 = args.config
        with tempfile.TemporaryDirectory() as td:
            from multiz_amd import synth as _synth
            spec = {"config": cfg0, "pairs": args.pairs or _synth.CONFIGS[cfg0]["pairs"], "first_pair": 0,
                    "seconds": args.cpu_seconds, "out": os.path.join(td, "cpu.npz")}
            json.dump(spec, open(os.path.join(td, "spec.json"), "w"))
            env = {k: v for k, v in os.environ.items() if not k.startswith("OMP_")}
            env["MZ_NO_TORCH"] = "1"
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", os.path.join(td, "spec.json")],
                               capture_output=True, text=True, env=env, cwd=ROOT)
            if p.returncode != 0:
                raise SystemExit("CPU-baseline leg failed:\n" + p.stderr[-2000:])
            cpu = json.loads(p.stdout.strip().splitlines()[-1])
            z = np.load(spec["out"])
            cpu_arrays = (z["idx"], z["om"], z["hs"])

    import numpy as np
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path in the product)")
    # MZ_BENCH_SHARE_GPU=1 (development only): every rank on GPU 0 with a gloo group, to exercise the N > 1 control
    # flow on a one-GPU box; the number it prints is not a scaling measurement
    share = world > 1 and os.environ.get("MZ_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    red = torch.device("cpu") if share else dev                 # where the exchange and the closing reductions live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # RCCL

    import multiz_amd as mz
    from multiz_amd import api, shard, synth

    if world > 1:
        # the ranks of a node share its cores: every rank's packing / assembling pool (mz_pool.c: 24 threads by default, for one process
        # on a 16-CPU quota) gets its share of one and a half times the CPUs this job may use
        os.environ.setdefault("MZ_HOST_THREADS", str(max(4, min(24, cpu_budget()[0] * 3 // 2 // world))))
    api.init(local)
    cfg = dict(synth.CONFIGS[args.config])
    pairs = args.pairs or cfg["pairs"]
    exchange = None
    if args.scatter and world > 1:
        # the list exists on rank 0 only: partition by band cells, one grouped RCCL send per peer, the shard's
        # tensors become the device batch where they land; the aligned shard goes back the same way (checked below)
        whole = synth.make_batch(pairs * world, cfg["K"], cfg["L"], cfg["mlo"], cfg["mhi"], cfg["radius"], indel=cfg.get("indel", 0)) if rank == 0 else None
        # on LINK IMAGES (include/mz_amd.h, mz_link_*): class nibbles + band steps out, a record per pair + 2-bit edit scripts back,
        # the root assembles the merged columns from its own pools; MZ_SCATTER_POOLS=1 sends the pools and brings the columns back
        torch.cuda.synchronize(dev)
        if os.environ.get("MZ_SCATTER_POOLS") == "1":
            t0 = time.perf_counter()
            tens, my_idx = shard.scatter_batch(whole, 0, red)
            torch.cuda.synchronize(dev)
            t_scatter = time.perf_counter() - t0
            db = mz.DevBatch.from_tensors(tens, device=dev)
            db.run()
            r = db.results_device()
            res_t = dict(om=r["om"], status=r["status"], off=r["offOut"], out=db.out[: int(r["totals"][2].item())])
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            sh = shard.gather_results(res_t, my_idx, pairs * world,
                                      (whole["K"].astype(np.int64) + whole["L"]) if rank == 0 else None, 0, red)
            torch.cuda.synchronize(dev)
            t_gather = time.perf_counter() - t0
            exchange = {"format": "pools", "scatter_s": round(t_scatter, 4), "gather_s": round(t_gather, 4), "pairs_this_rank": int(len(my_idx))}
        else:
            # the exchange in C (include/mz_shard.h): the library deals, packs, sends (RCCL: grouped ncclSend / ncclRecv), aligns where the
            # image lands and assembles on the root; this script only times the three calls
            t0 = time.perf_counter()
            share, my_idx = shard.scatter_link(whole, 0, red)
            torch.cuda.synchronize(dev)
            t_scatter = time.perf_counter() - t0
            t0 = time.perf_counter()
            res_l = shard.link_compute(share, device=dev)
            torch.cuda.synchronize(dev)
            t_align = time.perf_counter() - t0
            desc = share.desc
            h_image, h_exc = share.host_image()                   # (for the timed steps below: the share as a device-resident batch)
            image, exc = torch.from_numpy(h_image).to(dev), torch.from_numpy(h_exc).to(dev)
            t0 = time.perf_counter()
            sh = shard.gather_link(share, my_idx, whole, 0, red)
            torch.cuda.synchronize(dev)
            t_gather = time.perf_counter() - t0
            exchange = {"format": "link images, exchanged by the library (mz_shard_*: " + ("RCCL" if dist.get_backend() == "nccl" else "the group's send / recv") + ")",
                        "scatter_s": round(t_scatter, 4), "align_s": round(t_align, 4), "gather_s": round(t_gather, 4),
                        "pairs_this_rank": int(len(my_idx))}
            def check_sample(sh_, label):
                """every sampled pair's merged columns (the bytes the root assembled from ITS pools) against the compiled reference"""
                from oracle import mzoracle as mo
                idx = np.linspace(0, pairs * world - 1, num=min(400, pairs * world)).astype(np.int64)
                sub = shard.take(whole, idx)
                om_r, hs_r, _, bad_r = mo.ref_batch(sub, threads=8) if mo.have_reference() else mo.yama_batch(sub, variant=1, threads=8)
                mism = sum(int(sh_.om[i]) != int(om_r[k]) or
                           mo.fnv1a_np(sh_.cols(int(i)), mo.fnv1a_np(np.array([sh_.om[i]], dtype=np.int32).view(np.uint8))) != int(hs_r[k]) for k, i in enumerate(idx))
                assert bad_r == 0 and mism == 0, f"{label}: {mism} of {len(idx)} gathered pairs differ from the reference"
                return int(len(idx))
            if rank == 0:
                ex = dict(shard.last_exchange)
                exchange["bytes_per_pair"] = {"out": round(ex["up_bytes"] / ex["pairs"], 1), "back": round(ex["down_bytes"] / ex["pairs"], 1),
                                              "pools_out": round(sum(whole[k].nbytes for k in ("poolA", "poolB", "poolLB", "poolRB")) / ex["pairs"], 1),
                                              "columns_back": round(float((sh.om.astype(np.int64) * sh.widths).sum()) / ex["pairs"], 1)}
                if not args.no_cpu:
                    exchange["checked_pairs"] = check_sample(sh, "three phases")
            # ---- the same list through the exchange IN CHUNKS (mz_shard_run, round 6): the root packs chunk t+1 and assembles chunk t-3
            # while the transport moves chunk t down and chunk t-2's results up and every GPU aligns chunk t-1 -- one call on every rank
            sync_early = (lambda: (torch.cuda.synchronize(dev), dist.barrier()))
            sync_early()
            t0 = time.perf_counter()
            sh2, totals2, times2 = shard.run_sharded_chunks(whole, None, 0, red)
            torch.cuda.synchronize(dev)
            t_chunked = time.perf_counter() - t0
            tt = torch.tensor([t_chunked, times2["align_s"]], dtype=torch.float64, device=red)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            three = exchange["scatter_s"] + exchange["align_s"] + exchange["gather_s"]
            exchange["chunked"] = {"wall_s": round(float(tt[0].item()), 4), "chunks": int(times2["chunks"]), "steps": int(times2["steps"]),
                                   "root_sums_s": {k: round(times2[k], 4) for k in ("pack_s", "comm_s", "align_s", "assemble_s")},
                                   "align_s_max_over_ranks": round(float(tt[1].item()), 4),
                                   "three_phases_sum_s": round(three, 4),
                                   "wall_over_longest_phase": round(float(tt[0].item()) / max(exchange["scatter_s"], exchange["align_s"], exchange["gather_s"], 1e-9), 3),
                                   "totals": list(totals2)}
            if rank == 0:
                assert (sh2.status == 0).all(), "a pair came back without a result from the chunked exchange"
                if not args.no_cpu:
                    exchange["chunked"]["checked_pairs"] = check_sample(sh2, "chunked")
                sh2.release()
            tens = api.link_expand(desc, image, exc)              # the timed steps run on the shard as it arrived: the image, expanded in HBM
            db = mz.DevBatch.from_tensors(tens, device=dev)
        batch = {k: v.cpu().numpy() for k, v in tens.items()}                 # host copy for the byte accounting below
        if rank == 0:
            assert (sh.status == 0).all() and (sh.owner >= 0).all(), "a pair came back without a result"
            exchange["ranks_used"] = int(len(set(sh.owner.tolist())))
            if hasattr(sh, "release"):
                sh.release()
        del whole
        pairs_here = int(len(my_idx))
    else:
        batch = synth.make_batch(pairs, cfg["K"], cfg["L"], cfg["mlo"], cfg["mhi"], cfg["radius"], first_pair=rank * pairs, indel=cfg.get("indel", 0))
        db = mz.DevBatch(batch, device=dev)                      # inputs now resident in HBM
        pairs_here = pairs
    # rotating workspaces for the pipelined form: three, or one more than the batches whose DPs the library runs side by
    # side when a batch is too small to fill the GPU (c3: 5 000 pairs, c5: 1 000)
    nws = max(3, api.lib().mz_dev_pipeline_depth(pairs_here) + 1)
    ring = [db] + [db.alternate() for _ in range(nws - 1)]

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # per-kernel durations for the roofline: the batch with the phases serialised and a HIP event pair around
    # each, on the stream the kernels are launched on (outside the timed region; doubles as extra warm-up)
    kern_ms = np.zeros(4)
    rsteps = args.resident_steps if args.resident_steps > 0 else max(1, args.steps) if args.no_host else 100
    kreps = max(3, min(rsteps, 20))
    db.run()                                                     # (first use creates the library's helper streams)
    for _ in range(kreps):
        kern_ms += np.array(db.run(timed=True))
    kern_ms /= kreps
    sync_all()
    # production form for a stream of batches (mz_dev_run_async): the DPs run back to back; plan + prep of step
    # k+1 and the latency-bound traceback walk + emit of step k-1 run on two helper streams beside the DP of
    # step k, on three rotating workspaces.  Every step does all of its work; all K steps are complete before
    # the closing synchronisation.  The parity gate below checks what these pipelined steps left in EVERY
    # workspace they used.
    for i in range(args.warmup):
        ring[i % nws].run_async()
    db.wait()
    sync_all()
    t0 = time.perf_counter()
    for i in range(rsteps):
        ring[(args.warmup + i) % nws].run_async()
    db.wait()
    sync_all()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    workspaces = ring[:min(nws, args.warmup + rsteps)]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    res = db.results()
    failed = int((res["status"] != 0).sum())
    total_bytes, cells = algorithmic_bytes(batch, res["om"])
    assert cells == int(res["cells"].sum()), "device cell count differs from the host count"
    stats = torch.tensor([cells, pairs_here, failed], dtype=torch.float64, device=red)
    if world > 1:
        dist.all_reduce(stats)                                   # the batch-closing reduction
    all_cells, all_pairs, all_failed = (int(x) for x in stats.tolist())
    if all_failed:
        raise SystemExit(f"{all_failed} block pairs failed on the device -- number void")

    resident_ms = 1e3 * elapsed / rsteps
    resident_gcups = all_cells * rsteps / elapsed / 1e9
    dp_ms = float(kern_ms[1])
    roof_achieved = total_bytes / (dp_ms * 1e-3) / 1e9          # GB/s, algorithmic bytes over the DP kernel's time
    modes = np.bincount(res["mode"], minlength=14)
    # the DP kernel that took most of the batch's pairs: the row-parallel one (k_dp_row_big when the batch has blocks of four
    # rows or more), the lagged one, or the wavefront fallbacks
    big = bool((np.maximum(batch["K"], batch["L"]) >= 4).any())
    nrow = int(modes[5:9].sum())
    # (a launch that leaves the GPU at most 2 048 row-parallel waves takes the latency-tolerant build, k_dp_row_lat; at most a wave per SIMD:
    #  the build whose waves cannot share one, k_dp_row_solo: C5 as one batch)
    row_kernel = "k_dp_row_big" if big else "k_dp_row_solo" if 0 < nrow <= 1024 else "k_dp_row_lat" if nrow <= 2048 else "k_dp_row"
    dominant_kernel = max((nrow, row_kernel), (int(modes[11]), "k_dp_lag"),
                          (int(modes[:4].sum()), "k_dp"), (int(modes[4]), "k_dp_tstrip"), (int(modes[13]), "k_dp_duo"), (int(modes[9:11].sum() + modes[12]), "k_dp_wide"))[1]

    # ---- THE HEADLINE, the first column of SURVEY 8(d): host buffers in, malloc()ed merged columns out, through mz_yama_batch() -- pack,
    # H2D, kernels, D2H, assembly on the host; chunks pipelined.  Every rank runs it on its own batch (its own PCIe
    # link; the host's cores are shared); every call is bracketed by a barrier + synchronize on both sides and counts with the MAX over ranks.
    host_hash = None
    host = None
    if not args.no_host:
        jobs, outs = api.host_jobs(batch)
        for _ in range(max(2, args.warmup)):                     # warm-up: the rotating staging buffers and result blocks grow to size
            api.yama_batch_records(jobs, outs)
            host_om = outs["OM"].copy()
            api.free_outs(outs)
        # (the container's CPU quota is per 100 ms period: let the period the harness's own threads -- generator, checker --
        #  have drawn on run out before the library's host threads are timed)
        time.sleep(0.3)
        os.environ["MZ_TIMING"] = "0"
        # the K timed steps: K calls, EVERY one bracketed by barrier + synchronize on both sides (max over ranks), `host_gap_ms` (50) apart
        # OUTSIDE the brackets.  Why apart: the GPU boxes give a job 16 CPUs' worth of time per 100 ms (cpu.max "1600000 100000") and a
        # 50 000-pair call spends ~0.15 s of CPU on its 24 packing / assembling threads -- calls issued back to back draw 18 CPUs' worth and
        # the kernel parks the whole process for the rest of every period (`back_to_back` below has that rate and the cgroup's throttling
        # counters: C2 10.3 against 8.6 ms a call, C4 23.7 against 15.5 on the round's last box; on a box whose quota holds they are FASTER
        # back to back, 7.6 ms: neither the GPU nor the pool goes idle).  What is timed is one batch's wall time, SURVEY 8(d)'s definition.
        t_host = []
        thr0 = cpu_throttle()
        for _ in range(args.steps):
            sync_all()
            t = time.perf_counter()
            api.yama_batch_records(jobs, outs)
            torch.cuda.synchronize(dev)
            t_host.append(time.perf_counter() - t)
            api.free_outs(outs)
            if args.host_gap_ms > 0:
                time.sleep(args.host_gap_ms * 1e-3)
        thr1 = cpu_throttle()
        t_host = np.array(t_host, dtype=np.float64)
        if world > 1:                                            # a step ends when the slowest rank's call has
            tt = torch.from_numpy(t_host.copy()).to(red)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_host = tt.cpu().numpy()
        # ... and the same calls one after the other in ONE timed region: what a caller that streams batches gets under this box's CPU quota
        time.sleep(0.3)
        nb2b = max(3, min(args.steps, 25))
        sync_all()
        t = time.perf_counter()
        for _ in range(nb2b):
            api.yama_batch_records(jobs, outs)
            api.free_outs(outs)
        sync_all()
        t_b2b = (time.perf_counter() - t) / nb2b
        thr2 = cpu_throttle()
        if world > 1:
            tt = torch.tensor([t_b2b], dtype=torch.float64, device=red)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_b2b = float(tt.item())
        if cpu is not None:                                      # the host path's BYTES, hashed like the CPU leg's (checker code; untimed call)
            from oracle import mzoracle as mo
            api.yama_batch_records(jobs, outs)
            assert np.array_equal(outs["OM"], host_om)
            host_hash = mo.hash_cols(outs["cols"], outs["OM"], batch["K"] + batch["L"])
            api.free_outs(outs)
        assert np.array_equal(host_om, res["om"]), "host path and device-resident path disagree"
        t_med, t_sum = float(np.median(t_host)), float(t_host.sum())
        link = api.link_bytes(jobs)                              # what one call moves over PCIe, from the library's own accounting
        host = {"gcups": all_cells * args.steps / t_sum / 1e9, "ms": 1e3 * t_sum / args.steps, "gap_ms": args.host_gap_ms,
                "median_gcups": round(all_cells / t_med / 1e9, 2), "median_ms": round(1e3 * t_med, 2),
                "ms_all": [round(1e3 * x, 2) for x in t_host],
                "spread": {"calls": int(args.steps), "max_over_median": round(float(t_host.max()) / t_med, 3), "min_over_median": round(float(t_host.min()) / t_med, 3),
                           "calls_above_1.15_median": int((t_host > 1.15 * t_med).sum()),
                           "cgroup_throttled_during_the_calls": ({"periods": thr1[0] - thr0[0], "usec": thr1[1] - thr0[1]} if thr0 and thr1 else None)},
                "back_to_back": {"calls": nb2b, "ms_per_call": round(1e3 * t_b2b, 2), "gcups": round(all_cells / t_b2b / 1e9, 2),
                                 "cgroup_throttled_during_the_calls": ({"periods": thr2[0] - thr1[0], "usec": thr2[1] - thr1[1]} if thr1 and thr2 else None)},
                "link_bytes_per_pair": {"up": round(link[0] / len(jobs), 1), "down": round(link[1] / len(jobs), 1)}}

    out = {
        "metric": "GCUPS (DP cell updates/s) on yama block-pair merge",
        # value: K calls of mz_yama_batch(), host buffers in, merged columns out, transfers included (SURVEY 8d's wall time; --no-host:
        # the resident rate, and the line says so)
        "value": round(host["gcups"] if host else resident_gcups, 3), "unit": "GCUPS", "n_gpus": world,
        "steps": args.steps if host else rsteps, "warmup": args.warmup,
        "ms_per_step": round(host["ms"] if host else resident_ms, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "value_is": ("mz_yama_batch(): host buffers in, merged columns out -- packing, H2D, kernels, D2H and host-side assembly inside every timed call (SURVEY.md 8d)"
                     if host else "device-resident pipeline (--no-host)"),
        "config": {"workload": synth.describe(args.config, pairs),
                   "pairs_total": all_pairs, "band_cells_total": all_cells,
                   "parallelism": f"pairs sharded x{world}" + (" (list built on rank 0, RCCL scatter/gather)" if exchange else "")},
        # the second column of SURVEY 8(d): the batch already in HBM, the pipelined device-resident form (rounds 1-5 printed this as `value`)
        "value_resident": round(resident_gcups, 3), "resident_ms_per_step": round(resident_ms, 3), "resident_steps": rsteps,
        # ONE resident batch, plan -> DP -> walk -> emit one after the other (HIP events), nothing of a neighbouring batch beside it
        # (`value_resident` is the pipelined form: at C5 five batches abreast)
        "single_batch_gcups": round(cells / (float(kern_ms.sum()) * 1e-3) / 1e9, 1),
        "kernel_ms": {"plan": round(float(kern_ms[0]), 3), "dp": round(dp_ms, 3),
                      "walk": round(float(kern_ms[2]), 3), "emit": round(float(kern_ms[3]), 3)},
        "dp_modes": {str(m): int(c) for m, c in enumerate(modes) if c},
        # dominant kernel: k_dp_row (the DP; one launch per step).  achieved = algorithmic bytes of the
        # batch / its HIP-event time.  The kernel is VALU-issue bound, not HBM bound:
        # measured traffic (profiles/, separate --pmc passes) stays under 1 TB/s; `valu` prices the same launch
        # against the integer VALU issue roof (instructions x 4 cycles / (1024 SIMDs x 2.4 GHz)).
        "roofline": {"bound": "hbm", "kernel": dominant_kernel, "achieved": round(roof_achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(roof_achieved / HBM_PEAK_GBS, 5),
                     "traffic": None,
                     "bytes_per_cell": round(total_bytes / cells, 4), "algorithmic_bytes": total_bytes,
                     "dp_kernel_gcups": round(cells / (dp_ms * 1e-3) / 1e9, 1), "valu": None},
    }
    pmc = pmc_record(args.config, pairs, dominant_kernel) if not args.scatter else None
    if pmc:
        # The issue roof of the dominant DP launch: its VALU wave-instructions (SQ_INSTS_VALU), priced by opcode class, against the
        # SIMD cycles of THE SAME launch -- GRBM_GUI_ACTIVE of that kernel in the counter passes, summed over the 8 XCDs, hence / 8
        # -- so no clock estimate enters: frac = instructions x cycles each / (1024 SIMDs x the launch's own cycles) <= 1.
        cpi, mix_src, mix_stale, mix = isa_mix(dominant_kernel)
        clock = pmc.get("clock_ghz")
        out["roofline"]["traffic"] = pmc.get("traffic_bytes")
        v = {"insts_per_launch": pmc["SQ_INSTS_VALU"], "cycles_per_inst": round(cpi, 4), "simds": SIMDS,
             "priced_by": ({"source": mix_src, "stale": mix_stale, "loop_valu": mix["valu"], "loop_fast": mix["fast"],
                            "cycles": {"add_sub_and": VALU_CYCLES_FAST, "rest": VALU_CYCLES_REST}} if mix_src else
                           f"no profiles/r*_isa_hist.json record: every instruction at {VALU_CYCLES_REST} cycles")}
        if pmc.get("GRBM_GUI_ACTIVE"):
            cyc = pmc["GRBM_GUI_ACTIVE"] / 8.0
            v["launch_cycles"] = round(cyc)
            v["clock_ghz"] = clock
            v["frac"] = round(pmc["SQ_INSTS_VALU"] * cpi / (SIMDS * cyc), 4)
            assert v["frac"] <= 1.0, f"roofline.valu.frac {v['frac']} > 1: the opcode pricing is wrong"
            if v["frac"] > out["roofline"]["frac"]:
                # the roof the launch is nearest to: integer VALU issue, not HBM (frac / achieved / peak stay the HBM figures SURVEY 8d defines)
                out["roofline"]["bound"] = "valu"
            if clock:
                v["frac_at_this_runs_dp_time"] = round(pmc["SQ_INSTS_VALU"] * cpi / (SIMDS * clock * 1e9 * dp_ms * 1e-3), 4)
        out["roofline"]["valu"] = v
        out["roofline"]["pmc"] = {"source": pmc["source"], "stale": pmc["stale"], "kernel_avg_ms_in_stats_run": round(pmc.get("avg_ns", 0) / 1e6, 3)}
    if host:
        out["value_host"] = round(host["gcups"], 2)              # (the headline under its old name; rounds 3-5 printed the MEDIAN of the calls: host_median)
        out["host_ms_per_batch"] = round(host["ms"], 2)
        out["host_gap_ms"] = host["gap_ms"]
        out["host_median"] = {"gcups": host["median_gcups"], "ms": host["median_ms"]}
        out["host_ms_all"] = host["ms_all"]
        out["host_spread"] = host["spread"]
        out["back_to_back"] = host["back_to_back"]
        out["host_link_bytes_per_pair"] = host["link_bytes_per_pair"]
    if exchange:
        out["exchange"] = exchange
    if world > 1:
        # What makes a first multi-GPU run self-verifying: which physical device every rank held (PCI bus id + name from the
        # library's own context), every rank's own rate, and the spread.  distinct_devices < n_gpus means ranks shared a GPU
        # (the MZ_BENCH_SHARE_GPU development switch) and the number is NOT a scaling measurement.
        ident = api.device_identity(0).encode()[:127]
        mine = torch.zeros(128, dtype=torch.uint8)
        mine[: len(ident)] = torch.tensor(list(ident), dtype=torch.uint8)
        mine = mine.to(red)
        all_id = [torch.zeros(128, dtype=torch.uint8, device=red) for _ in range(world)]
        dist.all_gather(all_id, mine)
        mine_rate = torch.tensor([cells * rsteps / max(elapsed_local, 1e-9) / 1e9, float(cells)], dtype=torch.float64, device=red)
        all_rate = [torch.zeros(2, dtype=torch.float64, device=red) for _ in range(world)]
        dist.all_gather(all_rate, mine_rate)
        idents = [bytes(t.cpu().tolist()).rstrip(b"\0").decode() for t in all_id]
        rates = [round(float(t[0].item()), 2) for t in all_rate]
        out["devices"] = idents
        out["distinct_devices"] = len(set(idents))
        out["per_rank_gcups"] = rates                            # (of the resident pipeline: every rank's own GPU, nothing shared)
        out["rank_max_over_min"] = round(max(rates) / max(min(rates), 1e-9), 3)
        out["backend"] = "gloo (ranks share GPU 0: development switch, not a scaling measurement)" if share else "nccl (RCCL)"

    # ---- the text path: what mz_multiz / mz_roast run every merge through.  N = 1 only.
    if rank == 0 and world == 1 and not args.no_host and not args.no_pre:
        pre = pre_column(args.config, pairs, 0 if args.no_cpu else args.pre_check)
        out["value_pre"] = pre["v1"]["gcups"]
        out["value_pre_v0"] = pre["v0"]["gcups"]
        out["pre"] = pre

    # ---- parity gate against the CPU leg's hashes (rank 0, N=1 only)
    if cpu is not None:
        from oracle import mzoracle as mo
        idx, om, hs = cpu_arrays
        # parity gate: per-pair hash of (OM, merged column bytes), GPU vs CPU
        mism = 0
        W = batch["K"].astype(np.int64) + batch["L"]
        for w in workspaces:
            wres = w.results()
            host_out = w.out.cpu().numpy()
            for j, i in enumerate(idx):
                o0, m_ = int(wres["offOut"][i]), int(wres["om"][i])
                if m_ != int(om[j]) or mo.fnv1a_np(host_out[o0: o0 + m_ * int(W[i])],
                                                   mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8))) != int(hs[j]):
                    mism += 1
            del host_out
        hmism = 0
        if host_hash is not None:                                # the host-buffer path: OM and merged column BYTES of every sampled pair
            hmism = int((host_hash[idx] != hs).sum())
        if mism or hmism or cpu.pop("bad"):
            raise SystemExit(f"PARITY FAILURE: {mism} (device-resident) / {hmism} (host path) of {len(idx)} sampled pairs differ from the CPU reference -- number void")
        out["cpu_baseline"] = cpu
        out["parity"] = (f"ok: {len(idx)} sampled pairs x {len(workspaces)} workspaces bit-identical (OM + merged columns) "
                         f"to the CPU {cpu['kind']}" + ("; the host path's merged columns too" if host_hash is not None else ""))
        out["vs_cpu"] = {"value_over_measured": round(out["value"] / cpu["value"], 1),
                         "value_over_socket_linear": round(out["value"] / cpu["socket_linear"], 1),
                         "value_resident_over_measured": round(out["value_resident"] / cpu["value"], 1),
                         "value_resident_over_socket_linear": round(out["value_resident"] / cpu["socket_linear"], 1)}

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
