"""A guide-tree-scale run of the in-process tree driver: a balanced tree of <leaves> species (default 30: BASELINE configs[3]'s
tree), <blocks> blocks per pairwise input file, every merge of every level through mz_roast on one GPU -- with MZ_TIMING=1 the
library prints one JSON line per alignment batch (merges, band cells, seconds, bytes over the link) and the driver its phase times;
everything goes to stdout.  First the same tree at <check blocks> blocks per file against the stock roast with the stock CPU aligners,
block for block (the scale the stock chain finishes in a minute or two).
    python tests/tools/roast_big.py [leaves 30] [blocks 9000] [check blocks 600]"""
import json, os, subprocess, sys, tempfile, time
from multiprocessing import Pool
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs
REF = os.path.join(ROOT, "oracle", "_ref")
leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_big = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
n_chk = int(sys.argv[3]) if len(sys.argv) > 3 else 600
species = ["s%c%c1" % (97 + k // 26, 97 + k % 26) for k in range(leaves - 1)]


def balanced(names):
    if len(names) == 1:
        return names[0]
    h = (len(names) + 1) // 2
    return "(" + balanced(names[:h]) + " " + balanced(names[h:]) + ")"


tree = balanced(["ref"] + species)


def make_file(args):
    td, k, sp, n = args
    rng = np.random.default_rng(1000 + k)
    ref = inputs.ACGT[np.random.default_rng(1).integers(0, 4, size=n * 260 + 300)]
    f = f"ref.{sp}.sing.maf"
    inputs.write_maf(os.path.join(td, f), inputs.random_maf_file(rng, ref, n, 2, sp[:-1], stride=250 + (5 * k) % 40))
    return f


def make_inputs(n):
    td = tempfile.mkdtemp()
    t = time.perf_counter()
    with Pool(min(16, os.cpu_count() or 8)) as pool:
        files = pool.map(make_file, [(td, k, sp, n) for k, sp in enumerate(species)])
    print(f"{leaves - 1} input files of {n} blocks generated in {time.perf_counter() - t:.1f} s", flush=True)
    return td, files


def body(path):
    return [l for l in open(path).read().split("\n") if not l.startswith("#")]


def ours(td, files, out):
    t = time.perf_counter()
    q = subprocess.run([os.path.join(ROOT, "multiz_amd", "mz_roast"), "E=ref", tree] + files + [os.path.join(td, out)], cwd=td,
                       env=dict(os.environ, MZ_TIMING="1"), capture_output=True)
    dt = time.perf_counter() - t
    assert q.returncode == 0, q.stderr.decode()[-2000:]
    return dt, [l for l in q.stderr.decode().split("\n") if (l.startswith("mz_") or l.startswith("{\"mz_")) and "chunk" not in l]


print("tree:", tree)
# -- parity at the scale the stock chain can run
td, files = make_inputs(n_chk)
run = os.path.join(td, "stock"); os.makedirs(os.path.join(run, "bin")); os.makedirs(os.path.join(run, "tmp"))
for name, target in {"maf_project": "maf_project_ref", "multiz": "multiz_ref", "multic": "multic_ref"}.items():
    os.symlink(os.path.join(REF, target), os.path.join(run, "bin", name))
t = time.perf_counter()
subprocess.run([os.path.join(REF, "roast_ref"), "T=" + os.path.join(run, "tmp"), "E=ref", tree] + files + [os.path.join(run, "out.maf")],
               cwd=td, env=dict(os.environ, PATH=os.path.join(run, "bin") + os.pathsep + os.environ["PATH"]), check=True, capture_output=True, timeout=3000)
t_stock = time.perf_counter() - t
t_ours, lines = ours(td, files, "ours.maf")
want, got = body(os.path.join(run, "out.maf")), body(os.path.join(td, "ours.maf"))
merges = sum(json.loads(l)["mz_preyama_batch"]["merges"] for l in lines if l.startswith("{\"mz_preyama_batch"))
print(f"{leaves} leaves x {n_chk} blocks: {merges} merges, {sum(l.startswith('a score=') for l in want)} blocks out; stock roast + stock multiz (CPU) {t_stock:.2f} s, "
      f"mz_roast {t_ours:.2f} s, identical: {got == want}", flush=True)
assert got == want
# -- the large run
td, files = make_inputs(n_big)
for rep in range(2):
    t_ours, lines = ours(td, files, "ours.maf")
    recs = [json.loads(l)["mz_preyama_batch"] for l in lines if l.startswith("{\"mz_preyama_batch")]
    merges, cells = sum(r["merges"] for r in recs), sum(r["cells"] for r in recs)
    inside = sum(r["seconds"] for r in recs[1:])
    print(f"run {rep}: {leaves} leaves x {n_big} blocks: {merges} merges in {len(recs)} alignment batches, {cells / 1e9:.2f} G band cells, wall {t_ours:.2f} s "
          f"(process start to exit); batches after the first: {inside:.3f} s inside the library = {sum(r['cells'] for r in recs[1:]) / max(inside, 1e-9) / 1e9:.1f} GCUPS "
          f"over text in / rows out", flush=True)
print("\n".join(lines))
print("output blocks:", sum(l.startswith("a score=") for l in body(os.path.join(td, "ours.maf"))))
