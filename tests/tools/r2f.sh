cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2f
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2f/pytest.log 2>&1; tail -8 gpurun_out/r2f/pytest.log
for cfg in c2 c3 c4 c5; do
timeout 300 python bench.py --config $cfg --steps 20 --no-cpu --no-host 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['kernel_ms'], d['kernel_gcups'])"
done
timeout 600 python tests/tools/indel_bands.py 20000 10 2>&1 | tail -4
