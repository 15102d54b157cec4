cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2d
timeout 1200 python -m pytest tests/test_preyama.py tests/test_roast_inprocess.py tests/test_roast_integration.py tests/test_batched_multiz.py tests/test_batched_multic.py tests/test_dropin_multiz.py -m gpu -x -q > gpurun_out/r2d/pytest_drivers.log 2>&1; tail -25 gpurun_out/r2d/pytest_drivers.log
for cfg in c2 c4; do
timeout 300 python bench.py --config $cfg --steps 20 --no-cpu --no-host 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['kernel_ms'])"
done
bash tests/tools/f1_phases.sh 3 20000 2>&1 | tail -12
MZ_HOST_PREP=1 bash tests/tools/f1_phases.sh 3 20000 2>&1 | tail -12
