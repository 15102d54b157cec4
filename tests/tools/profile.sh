cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 280 python bench.py --steps 20 --warmup 3 --cpu-seconds 12 > gpurun_out/bench_r1h.json 2> gpurun_out/bench_r1h.err
tail -c 1500 gpurun_out/bench_r1h.json
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1h -- python3 bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/prof_r1h.log 2>&1
cat $(find gpurun_out/prof_r1h -name "*kernel_stats.csv" | head -1)
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" ; do
  d=gpurun_out/pmc_r1h_$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 250 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
  python3 - "$d" <<'PY'
import csv,glob,collections,sys
fs=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(float); cnt=collections.Counter()
for row in csv.DictReader(open(fs[0])):
    k=(row["Kernel_Name"].split("(")[0], row["Counter_Name"]); acc[k]+=float(row["Counter_Value"]); cnt[k]+=1
for k in sorted(acc):
    if k[0].startswith("k_"): print(k[0], k[1], round(acc[k]/cnt[k]), "calls", cnt[k])
PY
done
