"""Which shader clock did every launch of a DP kernel run at?  (HISTORY 9.12: the serial C5 DP takes 25.4 or 30.2 ms with the same binary.)
    python tests/tools/c5_clock.py <dir> [config] [kernel prefix] [host]   (on the GPU box)
One rocprofv3 pass with --kernel-trace --pmc GRBM_GUI_ACTIVE over `bench.py --config c5 --steps 12 --no-cpu --no-host`: per dispatch the
counter is the cycles the GPU was active during it, summed over the 8 XCDs; cycles / 8 / the dispatch's own duration = its clock."""
import csv, glob, os, subprocess, sys
out = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "c5"
pref = sys.argv[3] if len(sys.argv) > 3 else "k_dp_row"
py = os.path.realpath(sys.executable)
host = len(sys.argv) > 4 and sys.argv[4] == "host"          # ... or of the chunk pipeline's launches: tests/tools/hostpath.py (HOSTPATH_SLEEP as set)
cmd = [py, "tests/tools/hostpath.py", "0", cfg] if host else [py, "bench.py", "--config", cfg, "--steps", "12", "--warmup", "2", "--no-cpu", "--no-host"]
subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", "GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", out, "--"] + cmd,
               env=dict(os.environ, TMPDIR="/tmp", MZ_DP_STREAMS="1"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
dur = {}
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0])
rows = []
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Kernel_Name"].startswith(pref):
            d = dur.get(r["Dispatch_Id"])
            ns = d[0] if d else (int(r.get("End_Timestamp", 0)) - int(r.get("Start_Timestamp", 0)))
            if ns > 200_000:
                rows.append((d[1] if d else 0, r["Kernel_Name"].split("(")[0], ns, float(r["Counter_Value"])))
rows.sort()
t0 = rows[0][0] if rows else 0
for s, k, ns, cyc in rows:
    print(f"{(s - t0) / 1e6:9.1f} ms  {k:14s} {ns / 1e6:7.3f} ms  {cyc / 8 / ns:5.3f} GHz  ({cyc / 8 / 1e6:7.2f} M cycles)")
