cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
timeout 1200 python -m pytest tests/test_roast_inprocess.py tests/test_roast_integration.py tests/test_batched_multiz.py tests/test_batched_multic.py tests/test_dropin_multiz.py -m gpu -x -q > gpurun_out/r2c/pytest_drivers.log 2>&1; tail -25 gpurun_out/r2c/pytest_drivers.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2c/pytest_parity.log 2>&1; tail -5 gpurun_out/r2c/pytest_parity.log
for cfg in c2 c4; do
timeout 300 python bench.py --config $cfg --steps 20 --no-cpu --no-host 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['kernel_ms'])"
MZ_WALK=wave timeout 300 python bench.py --config $cfg --steps 20 --no-cpu --no-host 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg MZ_WALK=wave', d['value'], d['ms_per_step'], d['kernel_ms'])"
done
for cp in 3125 6250 12500; do MZ_CHUNK_PAIRS=$cp timeout 300 python tests/tools/hostpath.py 50000 c2 2>&1 | tail -1 | sed "s/^/chunk $cp: /"; done
