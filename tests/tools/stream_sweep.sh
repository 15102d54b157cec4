cd $GRAFT_REPO_ROOT
run() { python tests/tools/prepath.py c2 0 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pre v1', d['v1']['gcups'], d['v1']['ms_all'], 'v0', d['v0']['gcups'], d['v0']['ms_all'])"; python tests/tools/hostpath.py 0 c2 2>&1 | tail -2; }
echo "base"; run
echo "MZ_DP_FOUR=1"; MZ_DP_FOUR=1 run
