import os, sys, time, subprocess, tempfile
import numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(1)
ref = inputs.ACGT[rng.integers(0, 4, size=n * 260 + 300)]
d = tempfile.mkdtemp(prefix="f1_", dir="/tmp")
inputs.write_maf(d + "/a.maf", inputs.random_maf_file(rng, ref, n, 3, "p"))
inputs.write_maf(d + "/b.maf", inputs.random_maf_file(rng, ref, n, 3, "q", stride=300))
root = os.path.abspath('.')
outs = {}
for name, binary in (("reference binary (CPU)", root + "/oracle/_ref/multiz_ref"),
                     ("reference driver + libmzamd (one pair per call)", root + "/oracle/_ref/multiz_mzamd"),
                     ("batched driver mz_multiz", root + "/multiz_amd/mz_multiz")):
    for v in (1, 0):
        w = os.path.join(d, str(abs(hash(name)) % 100000) + str(v)); os.makedirs(w)
        t = time.perf_counter()
        p = subprocess.run([binary, "../a.maf", "../b.maf", str(v), "u1", "u2"], capture_output=True, cwd=w)
        dt = time.perf_counter() - t
        outs[(name, v)] = p.stdout
        print(f"{name:50s} v={v}: {dt:7.2f} s  rc={p.returncode} blocks={p.stdout.count(b'a score=')}", flush=True)
for v in (1, 0):
    ks = [k for k in outs if k[1] == v]
    print("v", v, "identical outputs:", all(outs[k] == outs[ks[0]] for k in ks))
