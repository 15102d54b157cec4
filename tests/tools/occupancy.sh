cd $GRAFT_REPO_ROOT
for d in 0 5120 12000 32000; do echo "dyn lds $d"; MZ_DYN_LDS=$d timeout 200 python tests/tools/modes.py 50000 1 2>&1 | grep "kernel ms"; done
