"""the input files of tests/tools/roast_big.py into a directory, and the tree on stdout   (python make_inputs.py <dir> [leaves 30] [blocks 9000])"""
import os, sys
import numpy as np
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs
td = sys.argv[1]; leaves = int(sys.argv[2]) if len(sys.argv) > 2 else 30; n = int(sys.argv[3]) if len(sys.argv) > 3 else 9000
species = ["s%c%c1" % (97 + k // 26, 97 + k % 26) for k in range(leaves - 1)]


def balanced(names):
    if len(names) == 1:
        return names[0]
    h = (len(names) + 1) // 2
    return "(" + balanced(names[:h]) + " " + balanced(names[h:]) + ")"


def make_file(args):
    k, sp = args
    rng = np.random.default_rng(1000 + k)
    ref = inputs.ACGT[np.random.default_rng(1).integers(0, 4, size=n * 260 + 300)]
    inputs.write_maf(os.path.join(td, f"ref.{sp}.sing.maf"), inputs.random_maf_file(rng, ref, n, 2, sp[:-1], stride=250 + (5 * k) % 40))


if __name__ == "__main__":
    os.makedirs(td, exist_ok=True)
    with Pool(min(8, os.cpu_count() or 8)) as pool:
        pool.map(make_file, list(enumerate(species)))
    print(balanced(["ref"] + species))
