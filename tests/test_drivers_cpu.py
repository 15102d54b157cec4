"""Host logic of the two batched drivers without a GPU: inputs on which no pair of blocks is merged (no overlap
for multiz; a common species in every overlapping pair for multic) never reach yama(), so mz_multiz / mz_multic
run to the end on a machine without a HIP device -- reader, list walk, pair enumeration, unused-part printing,
command line -- and must match the stock binaries byte for byte.  A run that does need the GPU must fail loudly."""
import os
import subprocess

import numpy as np
import pytest

import inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
BINS = {"multiz": (os.path.join(REF, "multiz_ref"), os.path.join(ROOT, "multiz_amd", "mz_multiz")),
        "multic": (os.path.join(REF, "multic_ref"), os.path.join(ROOT, "multiz_amd", "mz_multic"))}

pytestmark = pytest.mark.skipif(not all(os.path.exists(p) for pair in BINS.values() for p in pair), reason="binaries not built")


def run(binary, args, workdir, tag, outs):
    d = os.path.join(workdir, tag)
    os.makedirs(d)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")      # no device even on a GPU box
    p = subprocess.run([binary] + args, capture_output=True, timeout=300, cwd=d, env=env)
    return (p.stdout, p.returncode, p.stderr) + tuple(open(os.path.join(d, o), "rb").read() for o in outs if os.path.exists(os.path.join(d, o)))


def both(tmp_path, prog, args, outs=()):
    want = run(BINS[prog][0], args, str(tmp_path), "ref", outs)
    got = run(BINS[prog][1], args, str(tmp_path), "ours", outs)
    assert got == want
    return want


@pytest.mark.parametrize("v", [1, 0])
def test_multiz_without_overlaps(tmp_path, v):
    # file 2's blocks lie in the gaps between file 1's: nothing to merge, everything goes to out1 / out2 (or stdout)
    rng = np.random.default_rng(11 + v)
    ref = inputs.ACGT[rng.integers(0, 4, size=12 * 400 + 600)]
    b1 = inputs.random_maf_file(rng, ref, 12, 3, "p", stride=400, blen=(80, 150))
    b2 = inputs.random_maf_file(rng, ref[200:], 12, 2, "q", stride=400, blen=(80, 150))
    for b in b2:
        b.rows[0].start += 200
    inputs.write_maf(str(tmp_path / "a.maf"), b1)
    inputs.write_maf(str(tmp_path / "b.maf"), b2)
    want = both(tmp_path, "multiz", ["../a.maf", "../b.maf", str(v), "u1", "u2"], ("u1", "u2"))
    assert want[1] == 0 and want[3].count(b"a score=") == len(b1) and want[4].count(b"a score=") == len(b2)
    d2 = tmp_path / "second"
    d2.mkdir()
    for f in ("a.maf", "b.maf"):
        os.link(str(tmp_path / f), str(d2 / f))
    both(d2, "multiz", ["M=100", "../a.maf", "../b.maf", str(v), "all"])


@pytest.mark.parametrize("v", [1, 0])
def test_multic_when_every_pair_shares_a_species(tmp_path, v):
    rng = np.random.default_rng(21 + v)
    ref = inputs.ACGT[rng.integers(0, 4, size=10 * 300 + 400)]
    from test_batched_multic import overlapping_list
    inputs.write_maf(str(tmp_path / "a.maf"), overlapping_list(rng, ref, 10, 3, ("p", "p"), (260, 290)))
    inputs.write_maf(str(tmp_path / "b.maf"), overlapping_list(rng, ref, 10, 3, ("p", "p"), (300, 270)))
    want = both(tmp_path, "multic", ["../a.maf", "../b.maf", str(v), "u1", "u2"], ("u1", "u2"))
    assert want[1] == 0 and want[0].count(b"a score=") == 0 and want[3].count(b"a score=") == 20


def test_a_run_that_needs_the_gpu_fails_loudly_without_one(tmp_path):
    rng = np.random.default_rng(5)
    ref = inputs.ACGT[rng.integers(0, 4, size=6 * 260 + 300)]
    inputs.write_maf(str(tmp_path / "a.maf"), inputs.random_maf_file(rng, ref, 6, 2, "p"))
    inputs.write_maf(str(tmp_path / "b.maf"), inputs.random_maf_file(rng, ref, 6, 2, "q", stride=300))
    for prog in ("multiz", "multic"):
        got = run(BINS[prog][1], ["../a.maf", "../b.maf", "1"], str(tmp_path), prog, ())
        assert got[1] == 1 and b"no HIP device" in got[2], got[2][-300:]
