cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --config $1 --steps 20 --no-cpu --no-host 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"config\"][\"workload\"][:4], d[\"value\"], d[\"kernel_gcups\"], d[\"kernel_ms\"], d[\"dp_modes\"])"; }
echo default; run c2g; echo MZ_NO_TROLL=1; MZ_NO_TROLL=1 run c2g; echo MZ_NO_TSTRIP=1; MZ_NO_TSTRIP=1 run c2g
