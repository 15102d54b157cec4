# end-of-round measurements: tests, PMC records, bench lines of every configuration, the text path's profiles   (tests/tools/final.sh <tag>)
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
    cp gpurun_out/${TAG}/${TAG}_pmc.json $O/${TAG}_pmc_partial.json 2>/dev/null
fi
tail -10 $O/pmc_collect.txt
cp gpurun_out/${TAG}/${TAG}_kernel_stats_*.csv $O/ 2>/dev/null
python bench.py > $O/${TAG}_bench_c2.json 2> $O/bench_c2.err
for c in c3 c4 c5 c2i c4i c2w c2s c2g; do python bench.py --config $c --resident-steps 30 > $O/${TAG}_bench_$c.json 2> $O/bench_$c.err; done
# the text path (mz_preyama_batch): kernel stats + counters of k_pre / k_mid / k_fin beside the DP kernels, chunked as shipped and each kernel alone
bash tests/tools/profile_pre.sh c2 ${TAG}_pre > $O/${TAG}_pre_profile_c2.txt 2>&1
MZ_CHUNKS=1 MZ_CHUNK_PAIRS=1000000 bash tests/tools/profile_pre.sh c2 ${TAG}_pre_alone > $O/${TAG}_pre_alone_profile_c2.txt 2>&1
cp gpurun_out/${TAG}_pre_kernel_stats_c2_v*.csv gpurun_out/${TAG}_pre_alone_kernel_stats_c2_v*.csv $O/ 2>/dev/null
# the host paths' calls one by one: kernel timelines of one call each (band path, text path), 50 calls per configuration with the slow ones
# explained (tests/tools/stall_hunt.py), and the command-processor measurements the stream layout rests on (tests/tools/ub/chain.hip)
python tests/tools/timeline.py $O/tl_host host c2 > $O/${TAG}_timeline_host_c2.txt 2>&1
python tests/tools/timeline.py $O/tl_host_c2i host c2i > $O/${TAG}_timeline_host_c2i.txt 2>&1
python tests/tools/timeline.py $O/tl_pre pre c2 1 > $O/${TAG}_timeline_pre_c2_v1.txt 2>&1
rm -rf $O/tl_host $O/tl_host_c2i $O/tl_pre
for c in c2 c2i c3; do python tests/tools/stall_hunt.py $c 50; MZ_HEDGE_US=0 python tests/tools/stall_hunt.py $c 50 | sed 's/^/   (pieces never run twice) /'; done > $O/${TAG}_stall_hunt.txt 2>&1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tests/tools/ub/chain.hip -o /tmp/chain 2>/dev/null && { /tmp/chain; for a in "1 6 20 1 64" "2 6 10 1 64" "1 1 20 1 64" "2 2 10 1 64"; do GPU_MAX_HW_QUEUES=8 /tmp/chain $a | tail -3; done; } > $O/${TAG}_pipes.txt 2>&1
# round 6: the exchange in chunks beside the three phases (two ranks sharing this box's GPU over gloo), and what the serial C5 DP's two durations are
MZ_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --config c4 --pairs 30000 --scatter 2> $O/scatter.err | grep '^{' > $O/${TAG}_bench_c4_scatter_2ranks.json
{ echo "== MZ_SOLO=0: k_dp_row_lat (single-wave workgroups dealt out by the hardware)"; MZ_SOLO=0 python tests/tools/c5_where.py 2>&1 | grep serial; echo "== default: k_dp_row_solo"; python tests/tools/c5_where.py 2>&1 | grep serial; } > $O/${TAG}_c5_where.txt 2>&1
MZ_SOLO=0 python tests/tools/c5_clock.py $O/clk c5 > $O/${TAG}_c5_clock.txt 2>&1; rm -rf $O/clk
MZ_SOLO=0 python tests/tools/c5_state.py $O/state > $O/${TAG}_c5_state.txt 2>&1; rm -rf $O/state
python tests/tools/timeline.py $O/tl_c5 host c5 > $O/${TAG}_timeline_host_c5.txt 2>&1; rm -rf $O/tl_c5
python tests/tools/roast_bench.py > $O/${TAG}_roast_bench.txt 2>&1
# the guide-tree-scale run of the tree driver (30 leaves, ~1.9 M merges): per-batch JSON lines and phase times
timeout 900 python tests/tools/roast_big.py 30 9000 600 > $O/${TAG}_roast30.txt 2>&1
# ... and what its main thread does, every 250 us of wall time (tests/tools/hostprof: the sampler preloaded beside the library)
gcc -O2 -g -shared -fPIC tests/tools/hostprof/sampler.c -o /tmp/libsampler.so -ldl -lpthread && python tests/tools/hostprof/make_inputs.py /tmp/rin 30 9000 > /tmp/tree.txt && (cd /tmp/rin && \
  MZ_SAMPLER_WALL=250 MZ_SAMPLER_MATCH=libmzamd MZ_SAMPLER_OUT=/tmp/sampler.out LD_PRELOAD=/tmp/libsampler.so MZ_TIMING=1 $GRAFT_REPO_ROOT/multiz_amd/mz_roast E=ref "$(cat /tmp/tree.txt)" ref.*.sing.maf out.maf 2>&1 | grep "mz_roast" ) > $O/${TAG}_roast_host_profile.txt 2>&1
python tests/tools/hostprof/report.py multiz_amd/libmzamd.so /tmp/sampler.out 45 >> $O/${TAG}_roast_host_profile.txt 2>&1
python - "$O" "$TAG" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], sys.argv[2] + "_bench_*.json"))):
    try:
        d = json.load(open(f))
        r = d["roofline"]
        print(os.path.basename(f), "value (host)", d["value"], d["ms_per_step"], "resident", d.get("value_resident"), "median", (d.get("host_median") or {}).get("gcups"), "slow calls", (d.get("host_spread") or {}).get("calls_above_1.15_median"), "b2b", (d.get("back_to_back") or {}).get("gcups"),
              "pre", d.get("value_pre"), d.get("value_pre_v0"), "serial", d["single_batch_gcups"], d["kernel_ms"],
              "roof", r["frac"], "valu", (r.get("valu") or {}).get("frac"), "traffic", r.get("traffic"), "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("socket_linear"), d.get("vs_cpu"))
    except Exception as e:
        print(f, "unreadable", e)
PY
