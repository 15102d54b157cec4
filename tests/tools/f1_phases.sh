# phase times of a batched multiz run: tests/tools/f1_phases.sh [rows per file, default 3] [blocks, default 20000]
cd $GRAFT_REPO_ROOT
ROWS=${1:-3}; BLOCKS=${2:-20000}
python - $ROWS $BLOCKS <<'PY'
import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import inputs
rows, n = int(sys.argv[1]), int(sys.argv[2])
rng=np.random.default_rng(1)
ref=inputs.ACGT[rng.integers(0,4,size=n*260+300)]
inputs.write_maf('/tmp/a.maf', inputs.random_maf_file(rng, ref, n, rows, "p"))
inputs.write_maf('/tmp/b.maf', inputs.random_maf_file(rng, ref, n, rows, "q", stride=300))
PY
cd /tmp
for v in 1 0; do s=$(date +%s.%N); MZ_TIMING=1 $GRAFT_REPO_ROOT/multiz_amd/mz_multiz a.maf b.maf $v u1 u2 > /dev/null; e=$(date +%s.%N); echo "wall $(python3 -c "print(round($e - $s, 3))") s"; done
if [ -x $GRAFT_REPO_ROOT/oracle/_ref/multiz_ref ]; then s=$(date +%s.%N); $GRAFT_REPO_ROOT/oracle/_ref/multiz_ref a.maf b.maf 1 u1 u2 > /dev/null; e=$(date +%s.%N); echo "reference binary v=1 wall $(python3 -c "print(round($e - $s, 3))") s"; fi
