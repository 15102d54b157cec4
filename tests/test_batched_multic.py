"""The second caller of pre_yama(): the batched multic driver (multiz_amd/mz_multic = mz_multic_main() of
libmzamd.so; reference multic.c) against the stock binary (oracle/_ref/multic_ref) on the same MAF files --
block lists that are NOT single coverage, so one block is merged with several of the other file.  Identical
bytes on stdout, stderr, exit code and both leftover files, for v = 1 and v = 0, with and without [out1 out2],
R= / M= / s= flags, species clashes (pairs the driver must skip), several contigs."""
import os
import subprocess

import numpy as np
import pytest

import inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "multic_ref")
OUR_BIN = os.path.join(ROOT, "multiz_amd", "mz_multic")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(REF_BIN) and os.path.exists(OUR_BIN)), reason="binaries not built")]


def run(binary, args, workdir, tag, outs):
    d = os.path.join(workdir, tag)
    os.makedirs(d)
    p = subprocess.run([binary] + args, capture_output=True, timeout=600, cwd=d)
    assert p.returncode in (0, 1), p.stderr.decode()[-2000:]
    return (p.stdout, p.returncode, p.stderr) + tuple(open(os.path.join(d, o), "rb").read() for o in outs)


def both(tmp_path, args, outs=()):
    want = run(REF_BIN, args, str(tmp_path), "ref", outs)
    got = run(OUR_BIN, args, str(tmp_path), "gpu", outs)
    for w, g in zip(want, got):
        assert g == w
    return want


def overlapping_list(rng, ref, nblocks, rows, tags, strides):
    """several single-coverage lists over the same reference, merged and sorted by top-row start: the
    blocks of the result overlap one another (multic's input)"""
    blocks = []
    for tag, stride in zip(tags, strides):
        blocks += inputs.random_maf_file(rng, ref, nblocks, rows, tag, stride=stride)
    blocks.sort(key=lambda b: b.rows[0].start)
    return blocks


def two_files(tmp_path, seed, rows, nblocks=14, tags2=("q", "r")):
    rng = np.random.default_rng(seed)
    ref = inputs.ACGT[rng.integers(0, 4, size=nblocks * 300 + 400)]
    inputs.write_maf(str(tmp_path / "a.maf"), overlapping_list(rng, ref, nblocks, rows[0], ("p", "s"), (260, 290)))
    inputs.write_maf(str(tmp_path / "b.maf"), overlapping_list(rng, ref, nblocks, rows[1], tags2, (300, 270)))


@pytest.mark.parametrize("v", [1, 0])
@pytest.mark.parametrize("rows", [(2, 2), (3, 2), (2, 4)])
def test_matches_stock_binary_with_leftover_files(tmp_path, v, rows):
    two_files(tmp_path, 7 + 1000 * v + 10 * rows[0] + rows[1], rows)
    want = both(tmp_path, ["../a.maf", "../b.maf", str(v), "u1", "u2"], ("u1", "u2"))
    assert want[0].count(b"\na score=") >= 15               # more merges than blocks: one block meets several


@pytest.mark.parametrize("v", [1, 0])
def test_all_sinks_on_stdout(tmp_path, v):
    two_files(tmp_path, 600 + v, (3, 3))
    want = both(tmp_path, ["../a.maf", "../b.maf", str(v)])
    assert want[0].count(b"\na score=") >= 15 and b"##eof" not in want[0]


@pytest.mark.parametrize("flags", [["R=10"], ["M=60"], ["R=4", "M=12"], ["C=40", "R=20"], ["s=0"]])
def test_flags_and_trailing_words(tmp_path, flags):
    two_files(tmp_path, 950 + len(flags[0]), (2, 3), nblocks=10)
    both(tmp_path, flags + ["../a.maf", "../b.maf", "1", "u1", "u2", "nohead"], ("u1", "u2"))
    d2 = tmp_path / "second"
    d2.mkdir()
    for f in ("a.maf", "b.maf"):
        os.link(str(tmp_path / f), str(d2 / f))
    both(d2, flags + ["../a.maf", "../b.maf", "0", "all"])


def test_species_clashes_are_skipped(tmp_path):
    # the second file carries rows of a species the first file has too: those pairs are not merged (and with
    # v = 0 the reference row of file 1 does not count), everything else is
    two_files(tmp_path, 4321, (3, 3), tags2=("p", "r"))
    w1 = both(tmp_path, ["../a.maf", "../b.maf", "1", "u1", "u2"], ("u1", "u2"))
    d2 = tmp_path / "second"
    d2.mkdir()
    for f in ("a.maf", "b.maf"):
        os.link(str(tmp_path / f), str(d2 / f))
    w0 = both(d2, ["../a.maf", "../b.maf", "0", "u1", "u2"], ("u1", "u2"))
    assert w1[0].count(b"\na score=") >= 5 and w0[0].count(b"\na score=") >= 5


def test_paralog_categories(tmp_path):
    # s=1 / s=2 with copy=/amplifier= marks on the "a" lines: blocks of category 'a' are left out under s=2,
    # two blocks with a copy row each are not merged, and a species clash between unmarked blocks ends the stock
    # program ("No COLOR_ROW_NAME specified!") after the merges before it
    rng = np.random.default_rng(99)
    ref = inputs.ACGT[rng.integers(0, 4, size=10 * 300 + 400)]
    from oracle.mzoracle import format_block, score_range

    def write(path, blocks, mark):
        with open(path, "w") as f:
            f.write("##maf version=1 scoring=blastz\n")
            for k, b in enumerate(blocks):
                b.score = score_range(b, 0, b.textSize)
                lines = format_block(b).split("\n")
                if mark and k % 3 == 0:
                    lines[0] += " copy=1"
                if mark and k % 4 == 1:
                    lines[0] += " amplifier=0"
                f.write("\n".join(lines))
    write(str(tmp_path / "a.maf"), overlapping_list(rng, ref, 8, 3, ("p", "s"), (260, 290)), True)
    write(str(tmp_path / "b.maf"), overlapping_list(rng, ref, 8, 3, ("q", "r"), (300, 270)), True)
    for k, flags in enumerate((["s=1"], ["s=2"])):
        d = tmp_path / ("run%d" % k)
        d.mkdir()
        for f in ("a.maf", "b.maf"):
            os.link(str(tmp_path / f), str(d / f))
        both(d, flags + ["../a.maf", "../b.maf", "1", "u1", "u2"], ("u1", "u2"))
    # a clash between unmarked blocks under s=1
    write(str(tmp_path / "c.maf"), overlapping_list(rng, ref, 8, 3, ("p", "s"), (260, 290)), False)
    write(str(tmp_path / "d.maf"), overlapping_list(rng, ref, 8, 3, ("q", "p"), (300, 270)), False)
    d = tmp_path / "clash"
    d.mkdir()
    want = both(d, ["s=1", "../../c.maf", "../../d.maf", "1"])
    assert want[1] == 1 and b"COLOR_ROW_NAME" in want[2]


def test_several_contigs_and_bad_usage(tmp_path):
    rng = np.random.default_rng(515)
    blocks1, blocks2 = [], []
    for name, n1, n2 in (("ref.chrA", 6, 6), ("ref.chrB", 5, 0), ("ref.chrC", 5, 7)):
        ref = inputs.ACGT[rng.integers(0, 4, size=8 * 300 + 400)]
        for blocks, n, tags in ((blocks1, n1, ("p", "s")), (blocks2, n2, ("q", "r"))):
            bl = overlapping_list(rng, ref, n, 3, tags, (260, 290)) if n else []
            for b in bl:
                b.rows[0].src = name
            blocks.extend(bl)
    n2a = sum(1 for b in blocks2 if b.rows[0].src == "ref.chrA")
    blocks2 = blocks2[n2a:] + blocks2[:n2a]                 # chrC before chrA in file 2
    inputs.write_maf(str(tmp_path / "a.maf"), blocks1)
    inputs.write_maf(str(tmp_path / "b.maf"), blocks2)
    both(tmp_path, ["../a.maf", "../b.maf", "1", "u1", "u2"], ("u1", "u2"))
    d2 = tmp_path / "second"
    d2.mkdir()
    for f in ("a.maf", "b.maf"):
        os.link(str(tmp_path / f), str(d2 / f))
    both(d2, ["../a.maf", "../b.maf", "0"])
    for k, args in enumerate(([], ["../a.maf"], ["../a.maf", "../b.maf", "2"], ["R=-1", "../a.maf", "../b.maf", "1"])):
        d = tmp_path / ("bad%d" % k)
        d.mkdir()
        want = both(d, [a.replace("../", "../../") if k else a for a in args] if False else args)
        assert want[1] == 1
