# same-box A/B of the run-following walk's LDS window (WIN_G groups of 16 rows: 12 KB at 16, 6 KB at 8)
cd $GRAFT_REPO_ROOT/multiz_amd/csrc
OBJS="mz_host.o mz_scores.o mz_preyama.o mz_maf.o mz_synth.o mz_mafio.o mz_multiz.o mz_multic.o mz_project.o mz_roast.o"
for g in 4 8; do
  sed "s/^#define WIN_G .*/#define WIN_G $g/" kernels/walk.inc > /tmp/walk_$g.inc
  mkdir -p /tmp/k$g/kernels; cp kernels/*.inc /tmp/k$g/kernels/; cp /tmp/walk_$g.inc /tmp/k$g/kernels/walk.inc; cp mz_device.hip mz_device.h /tmp/k$g/; mkdir -p /tmp/include; cp ../../include/*.h /tmp/include/ 2>/dev/null
  sed -i 's#"../../include/mz_amd.h"#"/tmp/include/mz_amd.h"#' /tmp/k$g/mz_device.h
  (cd /tmp/k$g && /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -c mz_device.hip -o dev.o 2>/dev/null)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libmz_win$g.so /tmp/k$g/dev.o $OBJS -Wl,-rpath,/opt/rocm/lib -lgomp -lpthread
done
cd $GRAFT_REPO_ROOT
for rep in 1; do for g in 4 8; do for c in c2 c3 c4 c5; do
  MZ_LIB_PATH=/tmp/libmz_win$g.so python bench.py --config $c --steps 20 --no-cpu --no-host 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('WIN_G $g $c', d['value'], d['ms_per_step'])"
done; MZ_DP_STREAMS=2 MZ_LIB_PATH=/tmp/libmz_win$g.so python bench.py --config c4 --steps 20 --no-cpu --no-host 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('WIN_G $g c4 two abreast', d['value'], d['ms_per_step'])"; done; done
