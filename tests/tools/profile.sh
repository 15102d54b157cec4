# bench.py + rocprofv3 kernel stats + the separate --pmc passes behind profiles/<tag>_*   (tests/tools/profile.sh <tag> [config])
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r2}; CFG=${2:-c2}
O=gpurun_out/prof_$TAG; mkdir -p $O
timeout 400 python bench.py --config $CFG > $O/bench_$CFG.json 2> $O/bench_$CFG.err
tail -c 3500 $O/bench_$CFG.json
MZ_DP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$CFG -- python3 bench.py --config $CFG --steps 20 --warmup 3 --no-cpu --no-host > $O/stats_$CFG.log 2>&1
f=$(find $O/stats_$CFG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cat "$f" < /dev/null
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" ; do
  d=$O/pmc_${CFG}_$(echo $c | tr ' ' '_' | cut -c1-40)
  MZ_DP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu --no-host > /dev/null 2>&1
  python3 - "$d" <<'PY'
import csv,glob,collections,sys
fs=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
if not fs: sys.exit(0)
acc=collections.defaultdict(float); cnt=collections.Counter()
for row in csv.DictReader(open(fs[0])):
    k=(row["Kernel_Name"].split("(")[0], row["Counter_Name"]); acc[k]+=float(row["Counter_Value"]); cnt[k]+=1
for k in sorted(acc):
    if k[0].startswith("k_"): print(k[0], k[1], round(acc[k]/cnt[k]), "calls", cnt[k])
PY
done
