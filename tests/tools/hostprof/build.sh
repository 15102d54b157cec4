# the tree driver's host side over the stand-in aligner, for profiling without a GPU   (bash tests/tools/hostprof/build.sh <out dir> [extra cc flags])
# e.g.  build.sh scratch/hostprof -DMZ_STAGE_THREADS=8        then   <out dir>/roast_prof E=ref "<tree>" files... destination
set -e
R=$(cd $(dirname $0)/../../.. && pwd); O=$1; shift; mkdir -p $O
for f in mz_roast mz_multiz mz_multic mz_project mz_maf mz_mafio mz_scores mz_preyama; do
    gcc -O2 -g -fopenmp -fno-inline-functions-called-once -I/opt/rocm/include "$@" -c $R/multiz_amd/csrc/$f.c -o $O/$f.o
done
gcc -O2 -g -fopenmp -I$R/include "$@" -c $R/tests/tools/hostprof/fake_align.c -o $O/fake_align.o
gcc -O2 -g -I$R/include -c $R/tests/tools/hostprof/sampler.c -o $O/sampler.o
gcc -fopenmp -no-pie "$@" $O/*.o -o $O/roast_prof -ldl
