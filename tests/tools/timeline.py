"""Kernel + copy timeline of ONE call of a host path under rocprofv3 (kernel trace + memory-copy trace), as text.
    python tests/tools/timeline.py <dir> pre|host [config] [pre-v]     (on the GPU box; rocprofv3's csv files land in <dir>)
Prints every DP / k_pre / k_fin / plan kernel and every copy of the last but one call with start, end, queue, and the union of the
DP kernels' busy time."""
import csv, glob, os, subprocess, sys
out, what = sys.argv[1], sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else "c2"
v = sys.argv[4] if len(sys.argv) > 4 else "1"
py = os.path.realpath(sys.executable)
cmd = [py, "bench.py", "--mode", "pre", "--config", cfg, "--pre-v", v, "--steps", "6", "--warmup", "2"] if what == "pre" else \
      [py, "tests/tools/hostpath.py", "0", cfg]
subprocess.run(["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--output-format", "csv", "-d", out, "--"] + cmd,
               env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
ev = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], "q" + r["Queue_Id"]))
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"].replace("MEMORY_COPY_", ""), "dma"))
ev.sort()
calls, cur = [], [ev[0]]
for a in ev[1:]:
    if a[0] - max(x[1] for x in cur) > 600_000:
        calls.append(cur); cur = [a]
    else:
        cur.append(a)
calls.append(cur)
big = [c for c in calls if len(c) > 100]
c = big[-2] if len(big) > 1 else calls[-1]
t0 = c[0][0]
skip = {"k_scan1", "k_scan2", "k_scan3", "k_fit", "k_rowprep", "k_emit_wide", "__amd_rocclr_fillBufferAligned", "k_plan_seg", "k_emit_long",
        "k_emit_long_count", "k_script_pack_long", "k_script_fin"}
for s, e, n, q in c:
    if n not in skip:
        print(f"{(s - t0) / 1e6:7.3f} {(e - t0) / 1e6:7.3f} {(e - s) / 1e3:7.1f}us {q:4s} {n}")
dp = sorted((s, e) for s, e, n, q in c if n.startswith("k_dp"))
busy, end = 0, 0
for s, e in dp:
    if e > end:
        busy += e - max(s, end); end = e
print(f"call: {(max(e for _, e, _, _ in c) - t0) / 1e6:.3f} ms on the GPU, DP kernels busy (union) {busy / 1e6:.3f} ms, sum {sum(e - s for s, e in dp) / 1e6:.3f} ms, {len(dp)} DP launches")
