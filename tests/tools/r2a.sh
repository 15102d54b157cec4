cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2a/pytest.log
tail -15 gpurun_out/r2a/pytest.log
timeout 300 python bench.py --steps 40 > gpurun_out/r2a/bench_c2.json 2> gpurun_out/r2a/bench_c2.err; tail -c 3000 gpurun_out/r2a/bench_c2.json; tail -5 gpurun_out/r2a/bench_c2.err
timeout 400 python bench.py --config c4 --steps 10 --cpu-seconds 10 > gpurun_out/r2a/bench_c4.json 2> gpurun_out/r2a/bench_c4.err; tail -c 3000 gpurun_out/r2a/bench_c4.json; tail -5 gpurun_out/r2a/bench_c4.err
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tests/tools/ub/ops.hip -o /tmp/ops 2>/dev/null && timeout 300 /tmp/ops > gpurun_out/r2a/ub_ops.txt 2>&1; head -70 gpurun_out/r2a/ub_ops.txt
