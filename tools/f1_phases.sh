cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import inputs
rng=np.random.default_rng(1); n=20000
ref=inputs.ACGT[rng.integers(0,4,size=n*260+300)]
inputs.write_maf('/tmp/a.maf', inputs.random_maf_file(rng, ref, n, 3, "p"))
inputs.write_maf('/tmp/b.maf', inputs.random_maf_file(rng, ref, n, 3, "q", stride=300))
PY
cd /tmp
for v in 1 0; do s=$(date +%s.%N); MZ_TIMING=1 $GRAFT_REPO_ROOT/multiz_amd/mz_multiz a.maf b.maf $v u1 u2 > /dev/null; e=$(date +%s.%N); echo "wall $(echo "$e - $s" | bc) s"; done
