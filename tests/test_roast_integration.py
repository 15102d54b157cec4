"""The tree driver on top: the reference's own `roast` (auto_mz.c, built unmodified into oracle/_ref/roast_ref)
reaches the aligner only by spawning `multiz` / `multic` (and `maf_project` between merges) from its PATH.  With
multiz_amd/mz_multiz and mz_multic there under the stock names, a whole reference-guided multiple alignment --
v = 0 and v = 1 merges along a four-species tree -- must come out block for block identical to the run with
the stock binaries.  (Comment lines carry the temp-file names with the process id and are left out.)"""
import os
import subprocess

import numpy as np
import pytest

import inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
NEED = [os.path.join(REF, x) for x in ("roast_ref", "maf_project_ref", "multiz_ref", "multic_ref")] + \
       [os.path.join(ROOT, "multiz_amd", x) for x in ("mz_multiz", "mz_multic")]

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not all(os.path.exists(p) for p in NEED), reason="binaries not built")]


def roast(tmp_path, tag, aligners, extra):
    run = tmp_path / tag
    (run / "bin").mkdir(parents=True)
    (run / "tmp").mkdir()
    os.symlink(os.path.join(REF, "maf_project_ref"), str(run / "bin" / "maf_project"))
    for name, target in aligners.items():
        os.symlink(target, str(run / "bin" / name))
    env = dict(os.environ, PATH=str(run / "bin") + os.pathsep + os.environ["PATH"])
    args = [os.path.join(REF, "roast_ref")] + extra + ["T=" + str(run / "tmp"), "E=ref", "((ref mouse1) (rat1 dog1))",
            "ref.mouse1.sing.maf", "ref.rat1.sing.maf", "ref.dog1.sing.maf", str(run / "out.maf")]
    p = subprocess.run(args, cwd=str(tmp_path), env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return [l for l in open(str(run / "out.maf")).read().split("\n") if not l.startswith("#")]


@pytest.mark.parametrize("extra", [[], ["P=multic"], ["R=12", "M=20"]])
def test_roast_with_gpu_aligners_on_the_path(tmp_path, extra):
    rng = np.random.default_rng(3 + len(extra))
    n = 25
    ref = inputs.ACGT[rng.integers(0, 4, size=n * 260 + 300)]
    for sp, stride in (("mouse", 260), ("rat", 250), ("dog", 270)):
        inputs.write_maf(str(tmp_path / f"ref.{sp}1.sing.maf"), inputs.random_maf_file(rng, ref, n, 2, sp, stride=stride))
    stock = roast(tmp_path, "stock", {"multiz": os.path.join(REF, "multiz_ref"), "multic": os.path.join(REF, "multic_ref")}, extra)
    ours = roast(tmp_path, "gpu", {"multiz": os.path.join(ROOT, "multiz_amd", "mz_multiz"),
                                   "multic": os.path.join(ROOT, "multiz_amd", "mz_multic")}, extra)
    assert sum(l.startswith("a score=") for l in stock) >= 20
    assert ours == stock
