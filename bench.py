#!/usr/bin/env python3
"""bench.py -- GCUPS (DP cell updates / s) of the yama block-pair merge on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one pass of the whole hot path -- validity/plan, banded DP with traceback bytes,
traceback walk, merged-column emit (reference mz_yama.c:58-313) -- over one batch of synthetic
block pairs that is already resident in HBM.  Workload at N=1: BASELINE.json configs[1]
("50k synthetic block pairs, 2+2 rows, ~1k x 1k cols, banded yama DP on 1 MI355X"); with N > 1
every rank runs its own 50k-pair shard of the same generator (weak scaling, no data-path
collective: block pairs are independent; one all-reduce of three scalars closes the batch).

Cells are band cells, counted exactly as the reference counts tback_size (mz_yama.c:60-66).
One JSON line on stdout (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# CPU-baseline threads (rank 0, N = 1 only): spread over the cores at once (must be set before any libgomp is loaded).
# Not with N > 1: binding pins each process's initial thread to the first place -- the SAME core for every rank.
if int(os.environ.get("WORLD_SIZE", "1")) == 1:
    os.environ.setdefault("OMP_PROC_BIND", "spread")

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# HBM bytes of one launch of the dominant kernel on the default c2 batch, from the PMC passes in profiles/
# (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs).  Measured, never estimated; valid for the default
# c2 batch only.
# k_dp_row on the c2 batch, profiles/r1h_pmc_summary.txt: FETCH_SIZE 355 553 KB (narrow coalesced reads are
# counted at one half on gfx950: x2) + WRITE_SIZE 2 438 496 KB
TRAFFIC_BYTES_PER_LAUNCH = (2 * 355553 + 2438496) * 1024


def algorithmic_bytes(batch, om):
    """K*M + L*N (columns in) + 8*(M+1) (LB,RB) + cells (1 B traceback per
    cell) + (M+N) (traceback read, upper bound) + (K+L)*OM (merged block out), summed over pairs."""
    K, L, M, N = (batch[k].astype(np.int64) for k in ("K", "L", "M", "N"))
    n_band = int(batch["offBand"][-1]) + int(M[-1]) + 1
    cells = int((batch["poolRB"][:n_band].astype(np.int64) - batch["poolLB"][:n_band] + 1).sum())
    total = int((K * M + L * N + 8 * (M + 1) + (M + N) + (K + L) * om.astype(np.int64)).sum()) + cells
    return total, cells


def fnv_rows(cols, om):
    from oracle import mzoracle as mo
    return mo.fnv1a_np(cols, mo.fnv1a_np(np.array([om], dtype=np.int32).view(np.uint8)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # (a step is ~5.4 ms; the pipeline's fill and drain cost ~2.5 ms per run)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=["c2", "c3"])
    ap.add_argument("--pairs", type=int, default=0, help="override pairs per GPU (default: the config's)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU-baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path in the product)")
    # MZ_BENCH_SHARE_GPU=1 (development only): every rank on GPU 0 with a gloo group, to exercise the N > 1 control
    # flow on a one-GPU box; the number it prints is not a scaling measurement
    share = world > 1 and os.environ.get("MZ_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    red = torch.device("cpu") if share else dev                 # where the closing reductions live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # RCCL

    import multiz_amd as mz
    from multiz_amd import api, synth

    api.init(local)
    cfg = dict(synth.CONFIGS[args.config])
    pairs = args.pairs or cfg["pairs"]
    batch = synth.make_batch(pairs, cfg["K"], cfg["L"], cfg["mlo"], cfg["mhi"], cfg["radius"],
                             first_pair=rank * pairs)
    db = mz.DevBatch(batch, device=dev)                          # inputs now resident in HBM
    ring = [db, db.alternate(), db.alternate()]                  # three rotating workspaces for the pipelined form

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # per-kernel durations for the roofline: the batch with the phases serialised and a HIP event pair around
    # each, on the stream the kernels are launched on (outside the timed region; doubles as extra warm-up)
    kern_ms = np.zeros(4)
    for _ in range(args.steps):
        kern_ms += np.array(db.run(timed=True))
    sync_all()
    # production form for a stream of batches (mz_dev_run_async): the DPs run back to back; plan + prep of step
    # k+1 and the latency-bound traceback walk + emit of step k-1 run on two helper streams beside the DP of
    # step k, on three rotating workspaces.  Every step does all of its work; all K steps are complete before
    # the closing synchronisation.  The parity gate below checks what these pipelined steps left in EVERY
    # workspace they used.
    for i in range(args.warmup):
        ring[i % 3].run_async()
    db.wait()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ring[(args.warmup + i) % 3].run_async()
    db.wait()
    sync_all()
    elapsed = time.perf_counter() - t0
    workspaces = ring[:min(3, args.warmup + args.steps)]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    res = db.results()
    failed = int((res["status"] != 0).sum())
    total_bytes, cells = algorithmic_bytes(batch, res["om"])
    assert cells == int(res["cells"].sum()), "device cell count differs from the host count"
    stats = torch.tensor([cells, pairs, failed], dtype=torch.float64, device=red)
    if world > 1:
        dist.all_reduce(stats)                                   # the batch-closing reduction
    all_cells, all_pairs, all_failed = (int(x) for x in stats.tolist())
    if all_failed:
        raise SystemExit(f"{all_failed} block pairs failed on the device -- number void")

    ms_per_step = 1e3 * elapsed / args.steps
    gcups = all_cells * args.steps / elapsed / 1e9
    dp_ms = kern_ms[1] / args.steps
    roof_achieved = total_bytes / (dp_ms * 1e-3) / 1e9          # GB/s, algorithmic bytes over the DP kernel's time

    out = {
        "metric": "GCUPS (DP cell updates/s) on yama block-pair merge",
        "value": round(gcups, 3), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {pairs} block pairs/GPU, {cfg['K']}+{cfg['L']} rows, "
                               f"M,N~U[{cfg['mlo']},{cfg['mhi']}], diag band R={cfg['radius']}",
                   "pairs_total": all_pairs, "band_cells_total": all_cells, "parallelism": f"pairs sharded x{world}"},
        "kernel_ms": {"plan": round(kern_ms[0] / args.steps, 3), "dp": round(dp_ms, 3),
                      "walk": round(kern_ms[2] / args.steps, 3), "emit": round(kern_ms[3] / args.steps, 3)},
        # dominant kernel: k_dp_row (the DP; one launch per step).  achieved = algorithmic bytes of the
        # batch / its HIP-event time.  The kernel is VALU-issue bound, not HBM bound:
        # measured traffic (profiles/, separate --pmc passes) stays under 1 TB/s.
        "roofline": {"bound": "hbm", "kernel": "k_dp_row", "achieved": round(roof_achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(roof_achieved / HBM_PEAK_GBS, 5), "traffic": TRAFFIC_BYTES_PER_LAUNCH if (args.config == "c2" and pairs == 50000) else None,
                     "bytes_per_cell": round(total_bytes / cells, 4), "algorithmic_bytes": total_bytes,
                     "dp_kernel_gcups": round(cells / (dp_ms * 1e-3) / 1e9, 1)},
    }

    # ---- CPU baseline + parity gate (rank 0, N=1 only).  Preferred: the REAL reference's yama()
    # (oracle/_ref/libref.so, built from /root/reference by oracle/Makefile, -O2) driven one pair per
    # thread; otherwise the oracle's faithful O(K*L)/cell restatement.  Both are checkers/baselines only.
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import mzoracle as mo
        cores = os.cpu_count() or 1          # (libgomp has pinned this thread by now; affinity would read 1)
        use_ref = mo.have_reference()
        run_cpu = (lambda bt: mo.ref_batch(bt, threads=cores)) if use_ref else (lambda bt: mo.yama_batch(bt, variant=0, threads=cores))
        rng = np.random.default_rng(12345)
        probe = synth.subset(batch, rng.choice(pairs, size=min(pairs, 4 * cores), replace=False))
        run_cpu(probe)                                            # warm-up: thread pool, page faults
        t = time.perf_counter()
        run_cpu(probe)
        per_pair = (time.perf_counter() - t) / len(probe["K"])
        nsample = int(max(cores, min(pairs, args.cpu_seconds / max(per_pair, 1e-7))))
        idx = np.sort(rng.choice(pairs, size=nsample, replace=False))
        sample = synth.subset(batch, idx)
        t = time.perf_counter()
        om, hs, ccells, bad = run_cpu(sample)
        cpu_s = time.perf_counter() - t
        # parity gate: per-pair hash of (OM, merged column bytes), GPU vs CPU
        mism = 0
        for w in workspaces:
            wres = w.results()
            host_out = w.out.cpu().numpy()
            for j, i in enumerate(idx):
                K, L = int(batch["K"][i]), int(batch["L"][i])
                o0, m_ = int(wres["offOut"][i]), int(wres["om"][i])
                if m_ != int(om[j]) or fnv_rows(host_out[o0: o0 + m_ * (K + L)], m_) != int(hs[j]):
                    mism += 1
            del host_out
        if mism or bad:
            raise SystemExit(f"PARITY FAILURE: {mism} of {nsample} sampled pairs differ from the CPU reference -- number void")
        what = ("reference yama() (oracle/_ref/libref.so, gcc -O2 -fcommon)" if use_ref
                else "oracle faithful O(K*L)/cell restatement (gcc -O2)")
        out["cpu_baseline"] = {"value": round(ccells / cpu_s / 1e9, 5), "unit": "GCUPS", "cores": cores,
                               "kind": "reference" if use_ref else "port",
                               "sample": f"{nsample} of the {pairs} pairs (seeded), {ccells} band cells in {cpu_s:.1f} s; "
                                         f"{what}, OpenMP one pair per thread on {cores} threads"}
        out["parity"] = f"ok: {nsample} sampled pairs x {len(workspaces)} workspaces bit-identical (OM + merged columns) to the CPU {out['cpu_baseline']['kind']}"

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
