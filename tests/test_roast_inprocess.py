"""The in-process tree driver multiz_amd/mz_roast: the reference's roast (auto_mz.c) with its chain
of child programs -- maf_project, multiz / multic, cp / mv / grep on temp files -- run inside one process on MAF text
in memory, sibling subtrees sharing their GPU batches.

CPU part: the projection step (mz_project.c) against the stock maf_project binary (built unmodified from
/root/reference by oracle/Makefile), on inputs that exercise what it does: reference row not on top, reference on the
minus strand (the block is reverse-complemented), several reference contigs, out-of-order blocks, neighbours that
continue one another (fused, with the seam squeezed), blocks without the reference (dropped).
GPU part: whole alignments, block for block against the stock roast with the stock aligners."""
import os
import subprocess

import numpy as np
import pytest

import inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
ROAST = os.path.join(ROOT, "multiz_amd", "mz_roast")
COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def _flip(block):
    """the same block written from the other strand"""
    from oracle.mzoracle import Block, Row
    rows = []
    for r in block.rows:
        rows.append(Row(src=r.src, start=r.srcSize - (r.start + r.size), size=r.size, strand="-" if r.strand == "+" else "+",
                        srcSize=r.srcSize, text=r.text.encode().translate(COMP)[::-1].decode()))
    return Block(rows=rows)


def _split(block, at):
    """two blocks that continue one another exactly (what maf_project fuses again)"""
    from oracle.mzoracle import Block, Row
    a, b = [], []
    for r in block.rows:
        t1, t2 = r.text[:at], r.text[at:]
        n1, n2 = sum(c != "-" for c in t1), sum(c != "-" for c in t2)
        if n1 == 0 or n2 == 0:
            return None
        a.append(Row(src=r.src, start=r.start, size=n1, strand=r.strand, srcSize=r.srcSize, text=t1))
        b.append(Row(src=r.src, start=r.start + n1, size=n2, strand=r.strand, srcSize=r.srcSize, text=t2))
    return Block(rows=a), Block(rows=b)


def _messy_maf(rng, path):
    from oracle.mzoracle import Block
    ref = inputs.ACGT[rng.integers(0, 4, size=30 * 260 + 300)]
    blocks = []
    for contig in ("chr1", "chr2"):
        bl = inputs.random_maf_file(rng, ref, 24, 3, "sp", stride=260)
        for b in bl:
            b.rows[0].src = "ref." + contig
        blocks += bl
    out = []
    for b in blocks:
        parts = [b]
        if rng.random() < 0.4:
            sp = _split(b, int(rng.integers(20, max(21, len(b.rows[0].text) - 20))))
            if sp:
                parts = list(sp)
        for p in parts:
            if rng.random() < 0.3:
                p = _flip(p)
            if rng.random() < 0.5:                         # the reference row somewhere below the top
                k = int(rng.integers(1, len(p.rows)))
                p = Block(rows=p.rows[1:k + 1] + [p.rows[0]] + p.rows[k + 1:])
            out.append(p)
    # a few blocks that do not name the reference at all
    for b in inputs.random_maf_file(rng, ref, 3, 2, "zz", stride=260):
        out.append(Block(rows=b.rows[1:] + b.rows[1:2]) if False else Block(rows=b.rows[1:]))
    order = rng.permutation(len(out))
    inputs.write_maf(path, [out[i] for i in order])


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "maf_project_ref")) and os.path.exists(ROAST)), reason="binaries not built")
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_projection_matches_the_stock_tool(tmp_path, seed):
    rng = np.random.default_rng(seed)
    path = str(tmp_path / "in.maf")
    _messy_maf(rng, path)
    want = subprocess.run([os.path.join(REF, "maf_project_ref"), path, "ref", str(tmp_path / "others.maf")],
                          capture_output=True, timeout=120)
    assert want.returncode == 0, want.stderr.decode()[-1000:]
    got = subprocess.run([ROAST, "--project", path, "ref"], capture_output=True, timeout=120)
    assert got.returncode == 0, got.stderr.decode()[-1000:]
    strip = lambda b: [l for l in b.decode().split("\n") if not l.startswith("#")]  # noqa: E731
    w, g = strip(want.stdout), strip(got.stdout)
    assert sum(l.startswith("a score=") for l in w) >= 30
    assert g == w
    # the fusion pass in its long-list form (the answers for all neighbouring pairs first, on all threads; by default only lists of
    # 20 000 blocks and more take it)
    got = subprocess.run([ROAST, "--project", path, "ref"], capture_output=True, timeout=120, env=dict(os.environ, MZ_FUSE_PARALLEL_MIN="2"))
    assert got.returncode == 0, got.stderr.decode()[-1000:]
    assert strip(got.stdout) == w


def test_tree_parse_errors_and_plan():
    if not os.path.exists(ROAST):
        pytest.skip("binary not built")
    p = subprocess.run([ROAST, "-", "E=ref", "((ref a) (b (c d)))", "x", "out.maf"], capture_output=True, timeout=60)
    assert p.returncode == 0
    plan = p.stdout.decode().strip().split("\n")
    assert len(plan) == 4 and plan[-1].startswith("node 3:") and plan[-1].endswith("5 species")
    for tree, msg in (("((ref a) (b c)", "too many '('"), ("(ref a))", "parse error"), ("(ref a, b)", "improper character")):
        p = subprocess.run([ROAST, "E=ref", tree, "x", "out.maf"], capture_output=True, timeout=60)
        assert p.returncode == 1 and msg in p.stderr.decode(), (tree, p.stderr)


NEED = [os.path.join(REF, x) for x in ("roast_ref", "maf_project_ref", "multiz_ref", "multic_ref")] + [ROAST]


def _stock_roast(tmp_path, tree, files, extra):
    run = tmp_path / "stock"
    (run / "bin").mkdir(parents=True)
    (run / "tmp").mkdir()
    for name in ("maf_project", "multiz", "multic"):
        os.symlink(os.path.join(REF, name + "_ref"), str(run / "bin" / name))
    env = dict(os.environ, PATH=str(run / "bin") + os.pathsep + os.environ["PATH"])
    args = [os.path.join(REF, "roast_ref")] + extra + ["T=" + str(run / "tmp"), "E=ref", tree] + files + [str(run / "out.maf")]
    p = subprocess.run(args, cwd=str(tmp_path), env=env, capture_output=True, timeout=1800)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return [l for l in open(str(run / "out.maf")).read().split("\n") if not l.startswith("#")]


@pytest.mark.gpu
@pytest.mark.skipif(not all(os.path.exists(p) for p in NEED), reason="binaries not built")
@pytest.mark.parametrize("tree,extra", [("((ref mouse1) (rat1 dog1))", []), ("((ref mouse1) (rat1 dog1))", ["P=multic"]),
                                        ("((ref mouse1) (rat1 dog1))", ["R=12", "M=20"]),
                                        ("(((ref mouse1) (rat1 dog1)) ((cow1 pig1) (cat1 (bat1 fox1))))", []),
                                        ("(mouse1 (rat1 (dog1 (cow1 ref))))", []),
                                        ("(((ref mouse1) (rat1 dog1)) ((cow1 pig1) (cat1 (bat1 fox1))))", ["taint=cat1"])])
def test_whole_alignment_matches_the_stock_roast(tmp_path, tree, extra):
    # ("taint=<species>": MZ_ROAST_TAINT, the driver's test hook -- the nodes above that leaf run on MAF text with the stock chain's line
    # filters, as they would for a species named "...maf...", and are handed lists by the subtrees below them)
    env_extra = {"MZ_ROAST_TAINT": e[6:] for e in extra if e.startswith("taint=")}
    extra = [e for e in extra if not e.startswith("taint=")]
    rng = np.random.default_rng(7 + len(tree) + len(extra))
    n = 25
    ref = inputs.ACGT[rng.integers(0, 4, size=n * 260 + 300)]
    species = [w.strip("()") for w in tree.replace("(", " ").replace(")", " ").split() if w.strip("()") != "ref"]
    files = []
    for k, sp in enumerate(species):
        f = f"ref.{sp}.sing.maf"
        inputs.write_maf(str(tmp_path / f), inputs.random_maf_file(rng, ref, n, 2, sp[:-1], stride=250 + 5 * k))
        files.append(f)
    want = _stock_roast(tmp_path, tree, files, extra)
    p = subprocess.run([ROAST] + extra + ["E=ref", tree] + files + [str(tmp_path / "ours.maf")], cwd=str(tmp_path),
                       capture_output=True, timeout=900, env=dict(os.environ, MZ_TIMING="1", **env_extra))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = [l for l in open(str(tmp_path / "ours.maf")).read().split("\n") if not l.startswith("#")]
    assert sum(l.startswith("a score=") for l in want) >= 20
    assert got == want
    assert not any(f.startswith("_MZ_") for f in os.listdir(str(tmp_path)))          # no temporary files
    # blocks go from node to node as lists (the default); MAF text between the nodes, as round 3 did, gives the same file
    p = subprocess.run([ROAST] + extra + ["E=ref", tree] + files + [str(tmp_path / "ours_text.maf")], cwd=str(tmp_path),
                       capture_output=True, timeout=900, env=dict(os.environ, MZ_ROAST_TEXT="1"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert open(str(tmp_path / "ours_text.maf")).read().replace("ours_text.maf", "ours.maf") == open(str(tmp_path / "ours.maf")).read()
