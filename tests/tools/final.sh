# end-of-round check: the whole GPU suite, smoke, the default bench line and the other configurations
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 2500 $O/bench_c2.json
for c in c3 c4 c5; do timeout 900 python bench.py --config $c --cpu-seconds 8 > $O/bench_$c.json 2> $O/bench_$c.err; python3 -c "import json; d=json.load(open('$O/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['kernel_gcups'], d.get('value_host'), d['cpu_baseline']['value'], d['parity'][:40])"; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-host > $O/stats_c2.log 2>&1
f=$(find $O/stats_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" < /dev/null
