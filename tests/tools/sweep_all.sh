# every randomised differential sweep of tests/tools/ with fresh seeds, one after the other (each under its own timeout); a summary line each.
#   bash tests/tools/sweep_all.sh <seed base>      (on the GPU box)
cd $GRAFT_REPO_ROOT
S=${1:-100}
run() { echo "== $*"; timeout 900 "$@" 2>&1 | tail -${TAILN:-2} | cut -c1-400; }
run python tests/tools/random_sweep.py $S $((S+6))
run python tests/tools/extreme_sweep.py $S $((S+3))
run python tests/tools/host_sweep.py $S $((S+3))
MZ_CHUNK_PAIRS=37 run python tests/tools/host_sweep.py $((S+3)) $((S+5))
run python tests/tools/preyama_sweep.py $S $((S+4))
MZ_CHUNK_PAIRS=29 run python tests/tools/preyama_sweep.py $((S+4)) $((S+6))
MZ_PRE_LDS=0 run python tests/tools/preyama_sweep.py $((S+6)) $((S+7))
run python tests/tools/lag_stress.py 6000 $S
LAG_STRESS_ROWS=20 run python tests/tools/lag_stress.py 3000 $((S+1))
run python tests/tools/strip_stress.py 6000 $S
run python tests/tools/strip_stress.py 6000 $((S+1))
MZ_TROLL=1 run python tests/tools/strip_stress.py 4000 $((S+2))
run python tests/tools/magnitudes.py
