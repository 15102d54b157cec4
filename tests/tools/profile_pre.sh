# kernel stats and counters of the text path (mz_preyama_batch): bash tests/tools/profile_pre.sh [config] [tag]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CFG=${1:-c2}; TAG=${2:-pre}
PY=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
for v in 1 0; do
  O=gpurun_out/${TAG}_v$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $PY bench.py --mode pre --config $CFG --pre-v $v --steps 10 --warmup 2 > $O/line.json 2> $O/stats.err
  cp $(find $O/stats -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats_${CFG}_v$v.csv
  cp $(find $O/stats -name "*kernel_trace.csv" | head -1) gpurun_out/${TAG}_kernel_trace_${CFG}_v$v.csv
  for g in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
    d=$O/pmc_$(echo $g | tr ' ' '_' | cut -c1-30)
    rocprofv3 --kernel-trace --pmc $g --output-format csv -d $d -- $PY bench.py --mode pre --config $CFG --pre-v $v --steps 2 --warmup 1 > /dev/null 2> $d.err
  done
  python3 - $O $v <<'PY'
import csv, glob, sys, collections
O, v = sys.argv[1], sys.argv[2]
acc, cnt = collections.defaultdict(float), collections.Counter()
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
names = sorted({k[0] for k in acc})
ctrs = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY",
         "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"]
print("v =", v, "per launch (averages):", " ".join(ctrs))
for n in names:
    print("%-28s" % n[:28], " ".join("%12.0f" % (acc[(n, c)] / max(cnt[(n, c)], 1)) for c in ctrs), "calls", cnt[(n, ctrs[0])])
PY
  cat $O/line.json; head -14 gpurun_out/${TAG}_kernel_stats_${CFG}_v$v.csv | cut -c1-140
done
