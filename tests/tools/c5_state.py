"""What differs between the two durations of the serial C5 DP launch (HISTORY 9.12: 25.4 or 30.2 ms, the same binary)?
    python tests/tools/c5_state.py <dir>          (on the GPU box)
One rocprofv3 --pmc pass per counter group over `bench.py --config c5 --steps 8 --no-cpu --no-host` (13 serial launches of k_dp_row_lat,
then the pipelined ones); per group the launches' durations and counters, averaged over the slow and over the fast launches."""
import csv, glob, os, subprocess, sys, collections
out = sys.argv[1]
py = os.path.realpath(sys.executable)
GROUPS = ["GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY",
          "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM", "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC",
          "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU", "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH",
          "TCC_HIT_sum TCC_MISS_sum", "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum", "SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_HITS SQC_DCACHE_MISSES"]
for gi, g in enumerate(GROUPS):
    d = os.path.join(out, f"g{gi}")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + g.split() + ["--output-format", "csv", "-d", d, "--",
                    py, "bench.py", "--config", "c5", "--steps", "8", "--warmup", "2", "--no-cpu", "--no-host"],
                   env=dict(os.environ, TMPDIR="/tmp", MZ_DP_STREAMS="1"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dur = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    per = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_dp_row_lat") and r["Dispatch_Id"] in dur:
                per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = sorted((dur[i][0], dur[i][1], c) for i, c in per.items() if dur[i][1] > 5_000_000)
    if not rows:
        print(f"group [{g}]: no data (counter not available?)"); continue
    cut = (min(r[1] for r in rows) + max(r[1] for r in rows)) / 2
    for name, sel in (("slow", [r for r in rows if r[1] > cut]), ("fast", [r for r in rows if r[1] <= cut])):
        if not sel: continue
        avg = {k: sum(r[2].get(k, 0.0) for r in sel) / len(sel) for k in sel[0][2]}
        print(f"group [{g}] {name}: {len(sel)} launches, {sum(r[1] for r in sel) / len(sel) / 1e6:.2f} ms: " + "  ".join(f"{k} {v:.4g}" for k, v in sorted(avg.items())))
    print("   order of durations (ms):", " ".join(f"{r[1] / 1e6:.1f}" for r in rows))
