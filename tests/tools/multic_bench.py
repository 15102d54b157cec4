"""batched mz_multic against the stock multic binary on two overlapping block lists:
    python tests/tools/multic_bench.py [blocks per single-coverage list, default 4000] [rows per block, default 3]"""
import os, sys, time, subprocess, tempfile
import numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import inputs
from test_batched_multic import overlapping_list
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(5)
ref = inputs.ACGT[rng.integers(0, 4, size=n * 300 + 400)]
d = tempfile.mkdtemp(prefix="mc_", dir="/tmp")
inputs.write_maf(d + "/a.maf", overlapping_list(rng, ref, n, rows, ("p", "s"), (260, 290)))
inputs.write_maf(d + "/b.maf", overlapping_list(rng, ref, n, rows, ("q", "r"), (300, 270)))
root = os.path.abspath('.')
outs = {}
for name, binary in (("reference multic (CPU)", root + "/oracle/_ref/multic_ref"), ("batched mz_multic", root + "/multiz_amd/mz_multic")):
    for v in (1, 0):
        w = os.path.join(d, name[:3] + str(v)); os.makedirs(w)
        t = time.perf_counter()
        p = subprocess.run([binary, "../a.maf", "../b.maf", str(v), "u1", "u2"], capture_output=True, cwd=w)
        dt = time.perf_counter() - t
        outs[(name, v)] = (p.stdout, open(w + "/u1", "rb").read(), open(w + "/u2", "rb").read())
        print(f"{name:28s} v={v}: {dt:7.2f} s  rc={p.returncode} merged blocks={p.stdout.count(b'a score=')}", flush=True)
for v in (1, 0):
    ks = [k for k in outs if k[1] == v]
    print("v", v, "identical outputs:", all(outs[k] == outs[ks[0]] for k in ks))
