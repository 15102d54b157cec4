"""Where do the slow calls of mz_yama_batch() lose their time?  Runs `reps` calls of a config with MZ_TIMING=2 (50 ms apart, as bench.py
does), and for every call slower than 1.15 x the median compares each chunk's host time stamps with the median call's:
    python tests/tools/stall_hunt.py [config] [reps]
Prints all call times, max / median, and per slow call the stamp that first fell behind (packed = a packing piece came late: a host
thread was off the CPU; sent / launched = a stage thread; plan_wait / result_wait = the GPU)."""
import json, os, subprocess, sys
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = sys.argv[2] if len(sys.argv) > 2 else "50"
env = dict(os.environ, MZ_TIMING="2", HOSTPATH_REPS=reps, HOSTPATH_SLEEP="0.05")
p = subprocess.run([sys.executable, "tests/tools/hostpath.py", "0", cfg], env=env, capture_output=True, text=True)
calls, cur = [], []
for line in p.stderr.splitlines() + p.stdout.splitlines():
    if line.startswith('{"mz_yama_batch_chunk"'):
        cur.append(json.loads(line))
    elif line.startswith('{"mz_yama_batch":'):
        calls.append((json.loads(line)["mz_yama_batch"]["seconds"] * 1e3, sorted(cur, key=lambda d: d["mz_yama_batch_chunk"]))); cur = []
calls = calls[2:]                                     # (the first calls grow the buffers)
ts = sorted(c[0] for c in calls)
med = ts[len(ts) // 2]
print(f"{cfg}: {len(calls)} calls, median {med:.2f} ms, min {ts[0]:.2f}, max {ts[-1]:.2f}, max/median {ts[-1] / med:.3f}")
print(" ".join(f"{c[0]:.2f}" for c in calls))
ref = min(calls, key=lambda c: abs(c[0] - med))[1]
keys = [("packed_ms", None), ("sent_ms", None), ("plan_wait_ms", 1), ("launched_ms", None), ("result_wait_ms", 1), ("assembled_ms", None)]
def stamp(d, k, i):
    return d[k] if i is None else d[k][i]
for t, chunks in calls:
    if t < 1.15 * med or len(chunks) != len(ref):
        continue
    first = None
    for d, r in zip(chunks, ref):
        for k, i in keys:
            late = stamp(d, k, i) - stamp(r, k, i)
            if late > 0.6 and first is None:
                first = (d["mz_yama_batch_chunk"], k, late, stamp(d, k, i))
    print(f"  slow call {t:.2f} ms: first stamp more than 0.6 ms behind the median call's: " + (f"chunk {first[0]} {first[1]} +{first[2]:.2f} ms (at {first[3]:.2f})" if first else "none (spread over the call)"))
