"""Drop-in check at the level the north star names: the REFERENCE's own multiz driver (multiz.c main loop,
maf.c reader/writer, multi_util.c helpers -- compiled from /root/reference by oracle/Makefile into
oracle/_ref/multiz_mzamd) linked against libmzamd.so for yama()/pre_yama()/init_scores70()/mafScoreRange(),
against the stock reference binary (oracle/_ref/multiz_ref) on the same MAF files: identical bytes on
stdout and in both leftover files.  Needs the prebuilt binaries (they travel with gpurun) and a GPU."""
import os
import subprocess

import numpy as np
import pytest

import inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "multiz_ref")
OUR_BIN = os.path.join(ROOT, "oracle", "_ref", "multiz_mzamd")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(REF_BIN) and os.path.exists(OUR_BIN)), reason="oracle/_ref binaries not built")]


def run(binary, f1, f2, v, workdir, tag):
    # the driver echoes its argv into the output header, so both runs get identical argument strings
    d = os.path.join(workdir, tag)
    os.makedirs(d)
    p = subprocess.run([binary, "../a.maf", "../b.maf", str(v), "u1", "u2"], capture_output=True, timeout=600, cwd=d)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout, open(os.path.join(d, "u1"), "rb").read(), open(os.path.join(d, "u2"), "rb").read()


@pytest.mark.parametrize("v", [1, 0])
@pytest.mark.parametrize("rows", [(2, 2), (3, 2), (2, 4)])
def test_reference_driver_over_gpu_library(tmp_path, v, rows):
    rng = np.random.default_rng(1000 * v + 10 * rows[0] + rows[1])
    ref = inputs.ACGT[rng.integers(0, 4, size=30 * 260 + 300)]
    f1, f2 = str(tmp_path / "a.maf"), str(tmp_path / "b.maf")
    inputs.write_maf(f1, inputs.random_maf_file(rng, ref, 30, rows[0], "p"))
    inputs.write_maf(f2, inputs.random_maf_file(rng, ref, 30, rows[1], "q", stride=300))
    want = run(REF_BIN, f1, f2, v, str(tmp_path), "ref")
    got = run(OUR_BIN, f1, f2, v, str(tmp_path), "gpu")
    assert want[0].count(b"\na score=") + want[0].startswith(b"a score=") >= 10     # the run really merged blocks
    assert got[0] == want[0]
    assert got[1] == want[1] and got[2] == want[2]
