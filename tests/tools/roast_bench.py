"""Wall time of a reference-guided multiple alignment three ways: the stock roast with the stock aligners (CPU), the stock
roast with the GPU aligners on its PATH (a process -- and a HIP start-up -- per merge), and the in-process driver
mz_roast (one process, sibling subtrees in shared GPU batches).  Outputs compared block for block.
    python tests/tools/roast_bench.py [blocks per file, default 3000]"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs
REF = os.path.join(ROOT, "oracle", "_ref")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
tree = "(((ref mouse1) (rat1 dog1)) ((cow1 pig1) (cat1 (bat1 fox1))))"
species = [w for w in tree.replace("(", " ").replace(")", " ").split() if w != "ref"]
td = tempfile.mkdtemp()
rng = np.random.default_rng(1)
ref = inputs.ACGT[rng.integers(0, 4, size=n * 260 + 300)]
files = []
for k, sp in enumerate(species):
    f = f"ref.{sp}.sing.maf"
    inputs.write_maf(os.path.join(td, f), inputs.random_maf_file(rng, ref, n, 2, sp[:-1], stride=250 + 5 * k))
    files.append(f)

def stock(tag, aligners):
    run = os.path.join(td, tag); os.makedirs(os.path.join(run, "bin")); os.makedirs(os.path.join(run, "tmp"))
    os.symlink(os.path.join(REF, "maf_project_ref"), os.path.join(run, "bin", "maf_project"))
    for name, target in aligners.items(): os.symlink(target, os.path.join(run, "bin", name))
    env = dict(os.environ, PATH=os.path.join(run, "bin") + os.pathsep + os.environ["PATH"])
    t = time.perf_counter()
    subprocess.run([os.path.join(REF, "roast_ref"), "T=" + os.path.join(run, "tmp"), "E=ref", tree] + files + [os.path.join(run, "out.maf")],
                   cwd=td, env=env, check=True, capture_output=True)
    return time.perf_counter() - t, [l for l in open(os.path.join(run, "out.maf")).read().split("\n") if not l.startswith("#")]

t_cpu, want = stock("cpu", {"multiz": os.path.join(REF, "multiz_ref"), "multic": os.path.join(REF, "multic_ref")})
t_path, got_path = stock("path", {"multiz": os.path.join(ROOT, "multiz_amd", "mz_multiz"), "multic": os.path.join(ROOT, "multiz_amd", "mz_multic")})
def ours(extra_env, out):
    best, last = None, None
    for _ in range(3):
        t = time.perf_counter()
        q = subprocess.run([os.path.join(ROOT, "multiz_amd", "mz_roast"), "E=ref", tree] + files + [os.path.join(td, out)], cwd=td,
                           env=dict(os.environ, MZ_TIMING="1", **extra_env), capture_output=True)
        dt = time.perf_counter() - t
        assert q.returncode == 0, q.stderr.decode()[-2000:]
        if best is None or dt < best: best, last = dt, q
    return best, last
t_host, p_host = ours({"MZ_HOST_PREP": "1"}, "ours_host.maf")
t_in, p = ours({}, "ours.maf")
got = [l for l in open(os.path.join(td, "ours.maf")).read().split("\n") if not l.startswith("#")]
print(f"{len(species)} species x {n} blocks, {sum(l.startswith('a score=') for l in want)} blocks out")
print(f"stock roast + stock multiz (CPU):          {t_cpu:7.2f} s")
print(f"stock roast + mz_multiz on its PATH:       {t_path:7.2f} s   identical: {got_path == want}")
got_host = [l for l in open(os.path.join(td, "ours_host.maf")).read().split("\n") if not l.startswith("#")]
print(f"mz_roast, host pre_yama stages (MZ_HOST_PREP=1): {t_host:7.2f} s   identical: {got_host == want}   (best of 3)")
print(f"mz_roast, device pre_yama stages (v = 0 too):   {t_in:7.2f} s   identical: {got == want}   (best of 3)")
print("\n".join(l for l in p.stderr.decode().split("\n") if (l.startswith("mz_") or l.startswith("{\"mz_")) and "chunk" not in l))
