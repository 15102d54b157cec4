# value_pre / value_host of C2 against the number of chunks a call is cut into (MZ_CHUNKS): bash tests/tools/chunk_sweep.sh
cd $GRAFT_REPO_ROOT
for c in 4 6 8 12 16 24; do
  echo "MZ_CHUNKS=$c"; MZ_CHUNKS=$c python tests/tools/prepath.py c2 0 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pre v1', d['v1']['gcups'], d['v1']['ms_all'], 'v0', d['v0']['gcups'], d['v0']['ms_all'])"
  MZ_CHUNKS=$c python tests/tools/hostpath.py 0 c2 2>&1 | tail -2
done
