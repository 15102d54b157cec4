cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_shard_gpu.py -m gpu -x -q -k "range or rows or shard or scatter" > gpurun_out/r2b/pytest.log 2>&1; tail -5 gpurun_out/r2b/pytest.log
# C4 pipelined form: where does the step time go?
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2b/trace_c4 -- python3 bench.py --config c4 --steps 6 --warmup 2 --no-cpu --no-host > gpurun_out/r2b/trace_c4.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r2b/trace_c4/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_")]
t0=min(int(r["Start_Timestamp"]) for r in rows)
for r in rows[-60:]:
    print(r["Kernel_Name"].split("(")[0], r.get("Queue_Id"), round((int(r["Start_Timestamp"])-t0)/1e3,1), round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,1))
PY
MZ_WALK=wave timeout 300 python bench.py --config c4 --steps 10 --no-cpu --no-host 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 MZ_WALK=wave', d['value'], d['ms_per_step'], d['kernel_ms'])"
MZ_WALK=direct timeout 300 python bench.py --config c4 --steps 10 --no-cpu --no-host 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 MZ_WALK=direct', d['value'], d['ms_per_step'], d['kernel_ms'])"
timeout 300 python bench.py --config c3 --steps 20 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'], d['kernel_ms'], d.get('value_host'))"
timeout 300 python bench.py --config c5 --steps 5 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', d['value'], d['ms_per_step'], d['kernel_ms'], d.get('value_host'))"
MZ_TIMING=1 timeout 300 python tests/tools/hostpath.py 50000 c2 2>&1 | tail -20
