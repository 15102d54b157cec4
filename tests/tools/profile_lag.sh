# kernel stats and counters of the DP kernels on the indel mix (tests/tools/indel_bands.py <pairs> <events>), DP kernels
# back to back (MZ_DP_SERIAL=1) so that a kernel's duration is its own; then the side-by-side form for the batch time
#   bash tests/tools/profile_lag.sh [events]          -> gpurun_out/lagprof/summary.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
EV=${1:-10}
O=gpurun_out/lagprof; mkdir -p $O
: > $O/summary.txt
for mode in side serial; do
  if [ $mode = serial ]; then export MZ_DP_SERIAL=1; else unset MZ_DP_SERIAL; fi
  echo "== $mode: $(timeout 300 python3 tests/tools/indel_bands.py 20000 $EV 3 nocheck 2>&1 < /dev/null | grep modes)" >> $O/summary.txt
done
echo "== wavefront only (MZ_NO_LAG=1): $(MZ_NO_LAG=1 timeout 300 python3 tests/tools/indel_bands.py 20000 $EV 3 nocheck 2>&1 < /dev/null | grep modes)" >> $O/summary.txt
export MZ_DP_SERIAL=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tests/tools/indel_bands.py 20000 $EV 3 nocheck > /dev/null 2>&1 < /dev/null
timeout 30 python3 - $O >> $O/summary.txt <<'PY'
import glob, sys
fs = glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True)
print("== kernel stats (DP kernels back to back)")
print("\n".join(open(fs[0]).read().split("\n")[:9]) if fs else "no stats file")
PY
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES"; do
  d=$O/pmc_$(echo $c | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 tests/tools/indel_bands.py 20000 $EV 3 nocheck > /dev/null 2>&1 < /dev/null
  timeout 60 python3 - "$d" >> $O/summary.txt <<'PY'
import csv, glob, collections, sys
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(float); cnt = collections.Counter()
for row in csv.DictReader(open(fs[0])):
    k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"]); acc[k] += float(row["Counter_Value"]); cnt[k] += 1
for k in sorted(acc):
    if k[0] in ("k_dp_lag", "k_dp", "k_dp_row"): print(k[0], k[1], round(acc[k] / cnt[k]), "per launch,", cnt[k], "launches")
PY
done
cat $O/summary.txt
