cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2h
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r2h/pytest.log 2>&1; tail -8 gpurun_out/r2h/pytest.log
for cfg in c2 c4 c3; do
timeout 300 python bench.py --config $cfg --steps 20 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['kernel_ms'], 'host', d.get('value_host'), d.get('host_ms_per_batch'))"
done
for cp in 0 3125 12500; do MZ_CHUNK_PAIRS=$cp MZ_TIMING=1 timeout 300 python tests/tools/hostpath.py 50000 c2 2>&1 | tail -3 | sed "s/^/chunk $cp: /"; done
timeout 300 python tests/tools/hostpath.py 150000 c2 2>&1 | tail -1
timeout 300 python tests/tools/single_call.py 2>&1 | tail -3
