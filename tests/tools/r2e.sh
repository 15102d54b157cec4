cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python - 3 20000 <<'PY'
import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import inputs
rows, n = int(sys.argv[1]), int(sys.argv[2])
rng=np.random.default_rng(1)
ref=inputs.ACGT[rng.integers(0,4,size=n*260+300)]
inputs.write_maf('/tmp/a.maf', inputs.random_maf_file(rng, ref, n, rows, "p"))
inputs.write_maf('/tmp/b.maf', inputs.random_maf_file(rng, ref, n, rows, "q", stride=300))
inputs.write_maf('/tmp/a10.maf', inputs.random_maf_file(rng, ref, 5000, 10, "p"))
inputs.write_maf('/tmp/b10.maf', inputs.random_maf_file(rng, ref, 5000, 10, "q", stride=300))
PY
cd /tmp
for f in "a.maf b.maf" "a10.maf b10.maf"; do
for hp in 0 1 0 1; do s=$(date +%s.%N); MZ_HOST_PREP=$hp MZ_TIMING=1 $GRAFT_REPO_ROOT/multiz_amd/mz_multiz $f 1 u1 u2 2> err.txt > out$hp.maf; e=$(date +%s.%N); echo "== $f MZ_HOST_PREP=$hp wall $(python3 -c "print(round($e - $s, 3))") s"; grep -v "chunk(" err.txt | tail -6; grep "chunk(" err.txt | tail -1; done
cmp out0.maf out1.maf && echo "outputs identical"
done
cd $GRAFT_REPO_ROOT
for w in default direct default direct; do
if [ $w = default ]; then unset MZ_WALK; else export MZ_WALK=$w; fi
timeout 300 python bench.py --config c2 --steps 40 --no-cpu --no-host 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 walk=$w', d['value'], d['ms_per_step'])"
done
