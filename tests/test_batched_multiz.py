"""the batched multiz driver (multiz_amd/mz_multiz = mz_multiz_main() of
libmzamd.so: own MAF reader, own walk over the block lists, every pre_yama() of the run aligned in one or two
GPU batches) against the stock reference binary (oracle/_ref/multiz_ref) on the same MAF files: identical
bytes on stdout and in both leftover files, for v = 1 and v = 0, with and without [out1 out2], with R= / M=
flags, several contigs, single-row blocks and contigs that only one file has."""
import os
import subprocess

import numpy as np
import pytest

import inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "multiz_ref")
OUR_BIN = os.path.join(ROOT, "multiz_amd", "mz_multiz")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(REF_BIN) and os.path.exists(OUR_BIN)), reason="binaries not built")]


def run(binary, args, workdir, tag, outs, env=None):
    # the driver echoes its argv into the output header, so both runs get identical argument strings
    d = os.path.join(workdir, tag)
    os.makedirs(d)
    p = subprocess.run([binary] + args, capture_output=True, timeout=600, cwd=d, env=env)
    assert p.returncode in (0, 1), p.stderr.decode()[-2000:]
    # (a run that yama() ends with a fatal message must fail the same way, after the same partial output)
    return (p.stdout, p.returncode, p.stderr) + tuple(open(os.path.join(d, o), "rb").read() for o in outs)


def both(tmp_path, args, outs=()):
    want = run(REF_BIN, args, str(tmp_path), "ref", outs)
    got = run(OUR_BIN, args, str(tmp_path), "gpu", outs)
    for w, g in zip(want, got):
        assert g == w
    # the list walk in pieces of a few blocks side by side (mz_multiz.c, walk_contig; by default only lists of thousands of blocks are cut)
    got = run(OUR_BIN, args, str(tmp_path), "gpu_pieces", outs, env=dict(os.environ, MZ_WALK_PIECE_MIN="6"))
    for w, g in zip(want, got):
        assert g == w
    return want


def two_files(tmp_path, seed, rows, nblocks=30, stride2=300):
    rng = np.random.default_rng(seed)
    ref = inputs.ACGT[rng.integers(0, 4, size=nblocks * 260 + 300)]
    inputs.write_maf(str(tmp_path / "a.maf"), inputs.random_maf_file(rng, ref, nblocks, rows[0], "p"))
    inputs.write_maf(str(tmp_path / "b.maf"), inputs.random_maf_file(rng, ref, nblocks, rows[1], "q", stride=stride2))


@pytest.mark.parametrize("v", [1, 0])
@pytest.mark.parametrize("rows", [(2, 2), (3, 2), (2, 4), (1, 3)])
def test_matches_stock_binary_with_leftover_files(tmp_path, v, rows):
    two_files(tmp_path, 77 + 1000 * v + 10 * rows[0] + rows[1], rows)
    want = both(tmp_path, ["../a.maf", "../b.maf", str(v), "u1", "u2"], ("u1", "u2"))
    if rows[0] > 1:
        assert want[0].count(b"\na score=") >= 10           # the run really merged blocks


@pytest.mark.parametrize("v", [1, 0])
def test_matches_stock_binary_all_sinks_on_stdout(tmp_path, v):
    # without [out1 out2] merged blocks and unused parts interleave on stdout: the order must be replayed
    two_files(tmp_path, 500 + v, (3, 3))
    want = both(tmp_path, ["../a.maf", "../b.maf", str(v)])
    assert want[0].count(b"\na score=") >= 20


@pytest.mark.parametrize("flags", [["R=10"], ["M=40"], ["R=5", "M=12"], ["R=7"], []])
def test_flags_and_trailing_words(tmp_path, flags):
    two_files(tmp_path, 900 + len(flags), (2, 3), nblocks=20)
    both(tmp_path, flags + ["../a.maf", "../b.maf", "1", "u1", "u2", "nohead"], ("u1", "u2"))
    both_dir = tmp_path / "second"
    both_dir.mkdir()
    for f in ("a.maf", "b.maf"):
        os.link(str(tmp_path / f), str(both_dir / f))
    both(both_dir, flags + ["../a.maf", "../b.maf", "0", "all"])


def test_several_contigs_and_orphans(tmp_path):
    # blocks of three reference contigs in file 1, two of them (in another order) plus a fourth in file 2
    rng = np.random.default_rng(4242)
    blocks1, blocks2 = [], []
    for name, n1, n2 in (("ref.chrA", 8, 8), ("ref.chrB", 6, 0), ("ref.chrC", 7, 9)):
        ref = inputs.ACGT[rng.integers(0, 4, size=10 * 260 + 300)]
        for blocks, n, tag in ((blocks1, n1, "p"), (blocks2, n2, "q")):
            bl = inputs.random_maf_file(rng, ref, n, 3, tag) if n else []
            for b in bl:
                b.rows[0].src = name
            blocks.extend(bl)
    ref = inputs.ACGT[rng.integers(0, 4, size=5 * 260 + 300)]
    extra = inputs.random_maf_file(rng, ref, 4, 2, "z")
    for b in extra:
        b.rows[0].src = "ref.chrD"
    blocks2 = blocks2[8:] + extra + blocks2[:8]            # chrC, chrD, chrA
    inputs.write_maf(str(tmp_path / "a.maf"), blocks1)
    inputs.write_maf(str(tmp_path / "b.maf"), blocks2)
    both(tmp_path, ["../a.maf", "../b.maf", "1", "u1", "u2"], ("u1", "u2"))
    d2 = tmp_path / "second"
    d2.mkdir()
    for f in ("a.maf", "b.maf"):
        os.link(str(tmp_path / f), str(d2 / f))
    both(d2, ["../a.maf", "../b.maf", "0"])


def test_runs_that_never_reach_the_gpu(tmp_path):
    # The driver starts the GPU in a background thread while it reads its inputs (mz_warm_start).  Runs that end before that
    # thread has finished -- files without a common contig, and a command line in error -- must still end cleanly, with the
    # stock binary's bytes and exit status, and never under a thread that is inside the HIP runtime.
    rng = np.random.default_rng(99)
    for name, f, tag in (("ref.chrA", "a.maf", "p"), ("ref.chrB", "b.maf", "q")):
        ref = inputs.ACGT[rng.integers(0, 4, size=6 * 260 + 300)]
        bl = inputs.random_maf_file(rng, ref, 5, 2, tag)
        for blk in bl:
            blk.rows[0].src = name
        inputs.write_maf(str(tmp_path / f), bl)
    for rep in range(5):                                   # (a race, if there is one, needs more than one try)
        d = tmp_path / f"run{rep}"
        d.mkdir()
        for f in ("a.maf", "b.maf"):
            os.link(str(tmp_path / f), str(d / f))
        both(d, ["../a.maf", "../b.maf", "1", "u1", "u2"], ("u1", "u2"))
    d = tmp_path / "usage"
    d.mkdir()
    want = run(REF_BIN, ["../a.maf"], str(d), "ref", ())
    got = run(OUR_BIN, ["../a.maf"], str(d), "gpu", ())
    assert got[1] == want[1] == 1 and got[0] == want[0]


def test_reader_oddities(tmp_path):
    # comment lines (echoed by the stock reader), i/e/q lines (skipped), amplifier=/copy= tags on the "a" line,
    # tabs and extra blanks, lowercase and N bases, a trailing block without a final blank line
    rng = np.random.default_rng(31)
    ref = inputs.ACGT[rng.integers(0, 4, size=12 * 260 + 300)]
    b1 = inputs.random_maf_file(rng, ref, 10, 3, "p")
    b2 = inputs.random_maf_file(rng, ref, 10, 3, "q", stride=300)
    from oracle.mzoracle import format_block, score_range

    def write(path, blocks, decorate):
        with open(path, "w") as f:
            f.write("##maf version=1 scoring=blastz\n# a comment right after the header\n")
            for k, b in enumerate(blocks):
                b.score = score_range(b, 0, b.textSize)
                text = format_block(b)
                if decorate:
                    lines = text.split("\n")
                    if k % 3 == 0:
                        lines[0] = lines[0] + " copy=1" if len(b.rows) > 1 else lines[0]
                    if k % 3 == 1 and len(b.rows) > 2:
                        lines[0] = lines[0] + "\tamplifier=2"
                    if k % 2 == 0 and len(b.rows) > 1:
                        lines.insert(2, "i " + b.rows[1].src + " N 0 C 0")
                        lines.insert(3, "q " + b.rows[1].src + " " + "9" * b.textSize)
                    if k % 4 == 2:
                        lines.insert(len(lines) - 2, "e dog.chr7 100 50 + 1000 I")
                    text = "\n".join(lines)
                    if k % 5 == 4:
                        text = "# comment between blocks " + str(k) + "\n" + text
                f.write(text)
            if not decorate:
                f.write("##eof maf\n")

    write(str(tmp_path / "a.maf"), b1, True)
    write(str(tmp_path / "b.maf"), b2, False)
    both(tmp_path, ["../a.maf", "../b.maf", "1", "u1", "u2"], ("u1", "u2"))
    d2 = tmp_path / "second"
    d2.mkdir()
    for f in ("a.maf", "b.maf"):
        os.link(str(tmp_path / f), str(d2 / f))
    both(d2, ["../b.maf", "../a.maf", "0"])
