# every kernel of the text path alone: the whole call as ONE chunk (MZ_CHUNK_PAIRS=<pairs>), rocprofv3 kernel stats
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
PY=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
for v in 1 0; do
  rm -rf gpurun_out/alone_v$v
  MZ_CHUNKS=1 MZ_CHUNK_PAIRS=1000000 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/alone_v$v -- $PY bench.py --mode pre --config ${1:-c2} --pre-v $v --steps 6 --warmup 2 2> /dev/null | cut -c1-230
  head -16 $(find gpurun_out/alone_v$v -name "*kernel_stats.csv" | head -1) | cut -c1-120
done
