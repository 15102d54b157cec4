cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  d=gpurun_out/pmc_c5_$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 bench.py --config c5 --steps 1 --warmup 1 --no-cpu --no-host > /dev/null 2>&1
  python3 - "$d" <<'PY'
import csv,glob,collections,sys
fs=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(float); cnt=collections.Counter()
for row in csv.DictReader(open(fs[0])):
    k=(row["Kernel_Name"].split("(")[0], row["Counter_Name"]); acc[k]+=float(row["Counter_Value"]); cnt[k]+=1
for k in sorted(acc):
    if k[0]=="k_dp_row": print(k[0], k[1], round(acc[k]/cnt[k]), "calls", cnt[k])
PY
done
