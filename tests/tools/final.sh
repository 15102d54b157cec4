# end-of-round measurements: tests, bench lines of every configuration, PMC records   (tests/tools/final.sh <tag>)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r4}
O=gpurun_out/${TAG}_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
# the counters first: bench.py reads profiles/<tag>_pmc.json (roofline.traffic / roofline.valu) and marks it stale when the device sources changed
# (published only when every pass of every configuration succeeded: a partial record must not become the roofline's source)
if timeout 2400 python tests/tools/pmc_collect.py ${TAG} ${PMC_CONFIGS:-c2 c3 c4 c5 c2i c4i c2w c2s c2g} > $O/pmc_collect.txt 2>&1; then
    cp gpurun_out/${TAG}/${TAG}_pmc.json profiles/${TAG}_pmc.json
else
    echo "pmc_collect.py failed or was incomplete: profiles/${TAG}_pmc.json NOT updated"
fi
tail -6 $O/pmc_collect.txt
python bench.py > $O/${TAG}_bench_c2.json 2> $O/bench_c2.err
for c in c3 c4 c5 c2i c4i c2w c2s; do python bench.py --config $c --steps 30 > $O/${TAG}_bench_$c.json 2> $O/bench_$c.err; done
python - "$O" "$TAG" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], sys.argv[2] + "_bench_*.json"))):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["value"], "host", d.get("value_host"), d.get("host_ms_per_batch"), "serial", d["kernel_gcups"], d["kernel_ms"], "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("socket_linear"), d.get("vs_cpu"))
    except Exception as e:
        print(f, "unreadable", e)
PY
