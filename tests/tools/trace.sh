cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace1 -- python3 bench.py --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/trace1/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if r["Kernel_Name"].startswith("k_")]
t0=min(int(r["Start_Timestamp"]) for r in rows)
# print the 2nd step's timeline (production form): find sequences
for r in rows[-36:]:
    print(r["Kernel_Name"].split("(")[0], r.get("Queue_Id"), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
PY
