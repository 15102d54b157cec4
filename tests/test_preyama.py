"""pre_yama(): the oracle's Python restatement against the golden vectors (CPU), and the product's C
implementation -- which runs its yama() calls on the GPU -- against both (gpu)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import inputs
from oracle import mzoracle as mo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "preyama_golden.json")))


def to_block(d):
    return None if d is None else mo.Block(rows=[mo.Row(**r) for r in d["rows"]], score=d["score"])


def same_block(a, b):
    if a is None or b is None:
        return a is None and b is None
    return (len(a.rows) == len(b.rows) and a.score == b.score and
            all((x.src, x.start, x.size, x.strand, x.srcSize, x.text) == (y.src, y.start, y.size, y.strand, y.srcSize, y.text)
                for x, y in zip(a.rows, b.rows)))


def test_oracle_preyama_matches_golden():
    assert len(GOLD) >= 50
    for i, g in enumerate(GOLD):
        got, _ = mo.pre_yama(to_block(g["a1"]), to_block(g["a2"]), g["beg"], g["end"], g["radius"], g["v"])
        assert same_block(got, to_block(g["out"])), (i, g["v"])


@pytest.mark.skipif(not mo.have_reference(), reason="oracle/_ref/libref.so not built")
def test_oracle_preyama_matches_reference_random():
    ref = mo.Reference()
    rng = np.random.default_rng(99)
    done = 0
    for _ in range(150):
        a1, a2, beg, end = inputs.random_block_pair(rng, int(rng.integers(1, 5)), int(rng.integers(1, 5)), int(rng.integers(60, 240)))
        if end - beg < 12:
            continue
        v, R = int(rng.integers(0, 2)), int(rng.choice([15, 30, 50]))
        try:
            mine, _ = mo.pre_yama(a1, a2, beg, end, R, v)
        except RuntimeError:
            continue
        assert same_block(mine, ref.pre_yama(a1, a2, beg, end, R, v))
        assert mo.score_range(a1, 0, a1.textSize) == ref.score_range(a1, 0, a1.textSize)
        done += 1
    assert done > 100


# ------------------------------------------------------------------ the product (GPU)

@pytest.fixture(scope="module")
def product():
    import multiz_amd as m
    m.api.init(0)
    lib = m.lib()
    lib.pre_yama.restype = C.POINTER(mo.MafAli)
    lib.pre_yama.argtypes = [C.POINTER(mo.MafAli), C.POINTER(mo.MafAli), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.mafAliFree.argtypes = [C.POINTER(C.POINTER(mo.MafAli))]
    lib.init_scores70()
    return lib


def product_pre_yama(lib, a1, a2, beg, end, R, v):
    keep = mo._Keep()
    c1, c2 = mo.block_to_c(a1, keep), mo.block_to_c(a2, keep)
    p = lib.pre_yama(C.byref(c1), C.byref(c2), beg, end, R, v, None)
    out = mo.block_from_c(p)
    if p:
        lib.mafAliFree(C.byref(p))
    return out


@pytest.mark.gpu
def test_product_preyama_matches_golden(product):
    for i, g in enumerate(GOLD):
        got = product_pre_yama(product, to_block(g["a1"]), to_block(g["a2"]), g["beg"], g["end"], g["radius"], g["v"])
        assert same_block(got, to_block(g["out"])), (i, g["v"])


@pytest.mark.gpu
def test_product_preyama_matches_oracle_random(product):
    rng = np.random.default_rng(31337)
    done = 0
    for _ in range(120):
        a1, a2, beg, end = inputs.random_block_pair(rng, int(rng.integers(1, 6)), int(rng.integers(1, 6)), int(rng.integers(60, 400)))
        if end - beg < 12:
            continue
        v, R = int(rng.integers(0, 2)), int(rng.choice([15, 30, 50]))
        try:
            want, _ = mo.pre_yama(a1, a2, beg, end, R, v)
        except RuntimeError:
            continue
        assert same_block(product_pre_yama(product, a1, a2, beg, end, R, v), want)
        done += 1
    assert done > 80


@pytest.mark.gpu
def test_product_helpers_match_oracle(product):
    # smooth / mafScoreRange / mafWrite formatting of the shim against the oracle's restatements
    rng = np.random.default_rng(8)
    product.smooth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    product.mafScoreRange.restype = C.c_double
    product.mafScoreRange.argtypes = [C.POINTER(mo.MafAli), C.c_int, C.c_int]
    for _ in range(50):
        M, N, R = int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.integers(0, 45))
        LB, RB = inputs.wander_band(rng, M, N)
        l, r = LB.astype(np.int32).copy(), RB.astype(np.int32).copy()
        product.smooth(l.ctypes.data, r.ctypes.data, M, N, R)
        wl, wr = mo.smooth(LB, RB, M, N, R)
        assert np.array_equal(l, wl) and np.array_equal(r, wr)
    a1, _, _, _ = inputs.random_block_pair(rng, 4, 2, 200)
    keep = mo._Keep()
    c1 = mo.block_to_c(a1, keep)
    assert product.mafScoreRange(C.byref(c1), 0, a1.textSize) == mo.score_range(a1, 0, a1.textSize)


def test_product_score_range_profile_and_literal():
    # The shim's mafScoreRange() (host C, no GPU involved) counts byte classes per column when the tables have the
    # reference's structure and walks all row pairs otherwise.  Both against the oracle's literal restatement: many
    # rows, odd bytes, sub-ranges with and without a previous column; then with a table that has no class structure.
    import multiz_amd as m
    lib = m.lib()
    lib.mafScoreRange.restype = C.c_double
    lib.mafScoreRange.argtypes = [C.POINTER(mo.MafAli), C.c_int, C.c_int]
    lib.init_scores70()
    rng = np.random.default_rng(31)
    blocks = []
    for nrows in (1, 2, 3, 7, 18, 40):
        a1, _, _, _ = inputs.random_block_pair(rng, nrows, 1, 160)
        for r in a1.rows[1:]:                                   # sprinkle bytes of the "other" class
            t = bytearray(r.text.encode())
            for k in rng.integers(0, len(t), size=6):
                if t[k] != ord("-"):
                    t[k] = int(rng.choice(list(b"NnXacgt")))
            r.text = t.decode()
        blocks.append(a1)
    def check(sc=None):
        for b in blocks:
            keep = mo._Keep()
            c = mo.block_to_c(b, keep)
            n = b.textSize
            for start, size in ((0, n), (0, 1), (1, n - 1), (n // 3, n // 2), (n - 1, 1)):
                assert lib.mafScoreRange(C.byref(c), start, size) == mo.score_range(b, start, size, sc), (len(b.rows), start, size)
    check()
    # a private table: HOXD70 with one asymmetric entry -- not constant on classes, so the literal loop must run
    sc = mo.scores70()
    flat = np.ctypeslib.as_array(sc.ss).reshape(128, 128).copy()
    flat[ord("A"), ord("c")] += 7
    rows = (C.POINTER(C.c_int) * 128)(*[C.cast(flat[i].ctypes.data, C.POINTER(C.c_int)) for i in range(128)])
    ss = C.POINTER(C.POINTER(C.c_int)).in_dll(lib, "ss")
    saved = C.cast(ss, C.c_void_p).value
    try:
        C.c_void_p.in_dll(lib, "ss").value = C.addressof(rows)
        for b in blocks:
            keep = mo._Keep()
            c = mo.block_to_c(b, keep)
            want = 0
            txt = [np.frombuffer(r.text.encode(), dtype=np.uint8).astype(np.int64) for r in b.rows]
            gop = np.array(list(sc.gop), dtype=np.int64)
            for p in range(len(txt)):
                for q in range(p + 1, len(txt)):
                    want += int(flat[txt[p], txt[q]].sum())
                    d = lambda a: (a == ord("-")).astype(np.int64)
                    want -= int(gop[(d(txt[p][:-1]) << 3) | (d(txt[q][:-1]) << 2) | (d(txt[p][1:]) << 1) | d(txt[q][1:])].sum())
            assert lib.mafScoreRange(C.byref(c), 0, b.textSize) == float(want), len(b.rows)
    finally:
        C.c_void_p.in_dll(lib, "ss").value = saved
        lib.init_scores70()


# ------------------------------------------------------------------ pre_yama2 (three blocks; no caller in the reference)

def _three_blocks(rng, nrows2, nrows3, length):
    """X and a diverged copy Y; a1 = their pairwise alignment, a2 = a block topped by X, a3 = a block topped by Y;
    returns (a1, a2, a3, beg1, end1, begN, endN) with both ends on columns of a1 that pair a base with a base"""
    X = inputs.ACGT[rng.integers(0, 4, size=length)]
    xs, ys, Y = [], [], []
    for p in range(length):
        u = rng.random()
        if u < 0.05:                                   # a base only X has
            xs.append(int(X[p])); ys.append(inputs.DASHB)
            continue
        if u < 0.10:                                   # bases only Y has, then the pair
            for _ in range(int(rng.integers(1, 4))):
                b = int(inputs.ACGT[rng.integers(0, 4)])
                xs.append(inputs.DASHB); ys.append(b); Y.append(b)
        b = int(X[p]) if rng.random() < 0.85 else int(inputs.ACGT[rng.integers(0, 4)])
        xs.append(int(X[p])); ys.append(b); Y.append(b)
    Y = np.array(Y, dtype=np.uint8)
    txt = lambda v: bytes(v).decode("ascii")          # noqa: E731
    a1 = mo.Block(rows=[mo.Row(src="xs.chr1", start=0, size=length, strand="+", srcSize=length + 1000, text=txt(xs)),
                        mo.Row(src="ys.chr1", start=0, size=len(Y), strand="+", srcSize=len(Y) + 1000, text=txt(ys))])
    a2 = inputs.random_maf_block(rng, X, 0, length, nrows2, "p")
    a2.rows[0].src = "xs.chr1"
    a3 = inputs.random_maf_block(rng, Y, 0, len(Y), nrows3, "q")
    a3.rows[0].src = "ys.chr1"
    both = [k for k in range(len(xs)) if xs[k] != inputs.DASHB and ys[k] != inputs.DASHB]
    cb, ce = both[int(rng.integers(0, len(both) // 3))], both[int(rng.integers(2 * len(both) // 3, len(both)))]
    pos = lambda row, col: sum(1 for ch in row[:col] if ch != inputs.DASHB)   # noqa: E731
    return a1, a2, a3, pos(xs, cb), pos(xs, ce), pos(ys, cb), pos(ys, ce)


@pytest.mark.gpu
@pytest.mark.skipif(not mo.have_reference(), reason="oracle/_ref/libref.so not built")
def test_product_preyama2_matches_reference(product, tmp_path):
    # connectionAgreement2() is the caller's (align_util.c); a stand-in that always agrees is loaded globally, so this
    # compares pre_yama2 itself: checks, slicing, the band from the pairwise alignment, yama() -- on the GPU for the
    # product -- and mafBuild(..., top = 1).  The reference is loaded from a private copy of libref.so AFTER the
    # stand-in: an instance loaded earlier in the session keeps calling its own connectionAgreement2(), with pws = NULL.
    import shutil
    stub = os.path.join(ROOT, "oracle", "libca2stub.so")
    if not os.path.exists(stub):
        pytest.skip("oracle/libca2stub.so not built")
    C.CDLL(stub, mode=C.RTLD_GLOBAL)
    private = str(tmp_path / "libref_private.so")
    shutil.copy(mo.REF_PATH, private)
    ref = mo.Reference(private)
    sig = [C.POINTER(mo.MafAli)] * 3 + [C.c_int] * 5 + [C.c_void_p]
    for lib in (ref.lib, product):
        lib.pre_yama2.restype = C.POINTER(mo.MafAli)
        lib.pre_yama2.argtypes = sig
    rng = np.random.default_rng(1234)
    done = 0
    for _ in range(40):
        a1, a2, a3, beg1, end1, begN, endN = _three_blocks(rng, int(rng.integers(1, 5)), int(rng.integers(1, 5)), int(rng.integers(80, 400)))
        R = int(rng.choice([15, 30, 50]))
        outs = []
        for lib in (ref.lib, product):
            keep = mo._Keep()
            c1, c2, c3 = (mo.block_to_c(x, keep) for x in (a1, a2, a3))
            p = lib.pre_yama2(C.byref(c1), C.byref(c2), C.byref(c3), beg1, end1, begN, endN, R, None)
            outs.append(mo.block_from_c(p))
        assert same_block(outs[1], outs[0])
        done += outs[0] is not None
    assert done >= 30


# ------------------------------------------------------------------ device-side prep / post

def _prejob(a1, a2, beg, end, radius):
    cb1, ce1 = mo.pos2col(a1.rows[0], beg, a1.textSize), mo.pos2col(a1.rows[0], end, a1.textSize)
    cb2, ce2 = mo.pos2col(a2.rows[0], beg, a2.textSize), mo.pos2col(a2.rows[0], end, a2.textSize)
    return ([r.text[cb1:ce1 + 1].encode() for r in a1.rows], [r.text[cb2:ce2 + 1].encode() for r in a2.rows], radius), cb1, cb2


def _assemble(res, a1, cb1, a2, cb2):
    """what the caller of mz_preyama_batch() still does (mafBuild, mz_preyama.c:46-70): bookkeeping of the rows --
    sources in order, starts advanced by the bases left of the slice, rows without a base dropped"""
    if res["null"] or res["rows"] is None:
        return None
    src = [(r, cb1) for r in a1.rows] + [(r, cb2) for r in a2.rows[1:]]
    rows = []
    for (r, cb), text, size in zip(src, res["rows"], res["size"]):
        if size == 0:
            continue
        rows.append(mo.Row(src=r.src, start=r.start + sum(ch != "-" for ch in r.text[:cb]), size=size, strand=r.strand,
                           srcSize=r.srcSize, text=text.decode()))
    return mo.Block(rows=rows, score=res["score"]) if rows else None


@pytest.mark.gpu
def test_device_prep_and_post_match_the_oracle_on_golden_blocks(product):
    # the 60 reference-generated block pairs, every one as a one-stage merge (v = 1): text in, text out through
    # mz_preyama_batch() -- rmColDash, band walk, smooth, yama, transposition, base counts and mafScoreRange on the GPU --
    # against the oracle's restatement of pre_yama (itself pinned to the compiled reference above)
    import multiz_amd as m
    jobs, meta = [], []
    for g in GOLD:
        a1, a2 = to_block(g["a1"]), to_block(g["a2"])
        j, cb1, cb2 = _prejob(a1, a2, g["beg"], g["end"], g["radius"])
        jobs.append(j); meta.append((a1, cb1, a2, cb2, g))
    res = m.preyama_batch(jobs)
    for r, (a1, cb1, a2, cb2, g) in zip(res, meta):
        want, _ = mo.pre_yama(a1, a2, g["beg"], g["end"], g["radius"], 1)
        assert r["status"] == 0
        assert same_block(_assemble(r, a1, cb1, a2, cb2), want)
        if g["v"] == 1:
            assert same_block(_assemble(r, a1, cb1, a2, cb2), to_block(g["out"]))       # the reference's own output


@pytest.mark.gpu
def test_device_prep_and_post_random_blocks(product):
    # random block pairs: many rows, dash-heavy second blocks (columns removed, rows left without a base), small
    # radii, long overlaps; one batch
    import multiz_amd as m
    rng = np.random.default_rng(2)
    jobs, meta = [], []
    while len(jobs) < 150:
        n1, n2 = int(rng.integers(1, 9)), int(rng.integers(2, 9))
        a1, a2, beg, end = inputs.random_block_pair(rng, n1, n2, int(rng.integers(60, 900)))
        if end - beg < 12:
            continue
        R = int(rng.choice([15, 30, 50]))
        try:
            want, _ = mo.pre_yama(a1, a2, beg, end, R, 1)
        except RuntimeError:
            continue
        j, cb1, cb2 = _prejob(a1, a2, beg, end, R)
        jobs.append(j); meta.append((a1, cb1, a2, cb2, want))
    res = m.preyama_batch(jobs)
    dropped = 0
    for r, (a1, cb1, a2, cb2, want) in zip(res, meta):
        assert r["status"] == 0
        got = _assemble(r, a1, cb1, a2, cb2)
        assert same_block(got, want)
        dropped += sum(s == 0 for s in r["size"]) if r["size"] else 0
    # a second block that is nothing but dashes under the overlap: pre_yama() returns NULL
    a1, a2, beg, end = inputs.random_block_pair(rng, 2, 2, 200)
    a2.rows[1].text = "-" * a2.textSize
    j, cb1, cb2 = _prejob(a1, a2, beg, end, 30)
    r = m.preyama_batch([j])[0]
    assert r["null"] and r["rows"] is None and mo.pre_yama(a1, a2, beg, end, 30, 1)[0] is None


# ------------------------------------------------------------------ two-stage merges (v = 0) on the device

def _v0_job(a1, a2, beg, end, radius):
    j, cb1, cb2 = _prejob(a1, a2, beg, end, radius)
    return j + (0,), cb1, cb2


@pytest.mark.gpu
def test_device_two_stage_merges_on_golden_blocks(product):
    # the 60 reference-generated block pairs as TWO-stage merges (v = 0, mz_preyama.c:265-336): rmColDash of both blocks,
    # first yama(), mapping() with the reference's two defects, the composed and smoothed band, second yama(), mafBuild
    # and mafScoreRange -- all on the GPU through mz_preyama_batch().  30 of the golden cases were generated by the
    # compiled reference with v = 0: those are compared with the reference's own output, all 60 with the oracle.
    import multiz_amd as m
    jobs, meta = [], []
    for g in GOLD:
        a1, a2 = to_block(g["a1"]), to_block(g["a2"])
        if len(a1.rows) < 2:
            continue
        j, cb1, cb2 = _v0_job(a1, a2, g["beg"], g["end"], g["radius"])
        jobs.append(j); meta.append((a1, cb1, a2, cb2, g))
    assert sum(g["v"] == 0 for *_, g in meta) >= 20
    res = m.preyama_batch(jobs)
    for r, (a1, cb1, a2, cb2, g) in zip(res, meta):
        want, _ = mo.pre_yama(a1, a2, g["beg"], g["end"], g["radius"], 0)
        assert r["status"] == 0 and r["null_code"] in (0, 1)
        assert same_block(_assemble(r, a1, cb1, a2, cb2), want), g.get("tag")
        if g["v"] == 0:
            assert same_block(_assemble(r, a1, cb1, a2, cb2), to_block(g["out"]))       # the reference's own output


@pytest.mark.gpu
def test_device_two_stage_merges_random_blocks(product):
    # random block pairs as two-stage merges, mixed with one-stage merges in the same call: dash-heavy blocks (columns
    # removed from EITHER block, which is where the two reference defects of SURVEY appendix A.6 act), rows left without
    # a base, small radii, long overlaps
    import multiz_amd as m
    rng = np.random.default_rng(20)
    jobs, meta = [], []
    nv0 = 0
    while nv0 < 170:
        n1, n2 = int(rng.integers(2, 9)), int(rng.integers(2, 9))
        a1, a2, beg, end = inputs.random_block_pair(rng, n1, n2, int(rng.integers(60, 900)))
        if end - beg < 12:
            continue
        R = int(rng.choice([15, 30, 50]))
        v = 0 if rng.random() < 0.8 else 1
        try:
            want, _ = mo.pre_yama(a1, a2, beg, end, R, v)
        except RuntimeError:
            continue
        j, cb1, cb2 = _prejob(a1, a2, beg, end, R)
        jobs.append(j + (v,)); meta.append((a1, cb1, a2, cb2, want, v))
        nv0 += v == 0
    res = m.preyama_batch(jobs)
    removed_a = 0
    for r, (a1, cb1, a2, cb2, want, v) in zip(res, meta):
        assert r["status"] == 0, (r["status"], r["stage"], v)
        assert same_block(_assemble(r, a1, cb1, a2, cb2), want), v
        removed_a += v == 0 and r["M"] < len(a1.rows[0].text[cb1:])
    assert removed_a > 5                                    # rmColDash took columns out of the first block too
    # the first block has nothing but its top row: pre_yama() returns NULL after writing a2's slice to fpw2 -- code 2, the caller's job
    a1, a2, beg, end = inputs.random_block_pair(rng, 1, 3, 200)
    j, cb1, cb2 = _prejob(a1, a2, beg, end, 30)
    r = m.preyama_batch([j + (0,)])[0]
    assert r["null_code"] == 2 and r["rows"] is None
    # nothing below the top row of the first block but dashes under the overlap: NULL
    a1, a2, beg, end = inputs.random_block_pair(rng, 2, 2, 200)
    a1.rows[1].text = "-" * a1.textSize
    j, cb1, cb2 = _prejob(a1, a2, beg, end, 30)
    r = m.preyama_batch([j + (0,)])[0]
    assert r["null_code"] == 1 and mo.pre_yama(a1, a2, beg, end, 30, 0)[0] is None


@pytest.mark.gpu
def test_preyama_batch_chunk_pipeline(product):
    # mz_preyama_batch() cuts a call into chunks that go through four stages on rotating buffer sets (mz_prebatch.c: packer, two
    # launchers, collector).  MZ_CHUNK_PAIRS=23 (read once per process, hence the child) makes 300 merges -- one- and two-stage
    # mixed, dash-heavy, NULL results in between -- thirteen chunks on the threaded pipeline; every block against the oracle
    import subprocess
    import sys
    env = dict(os.environ, MZ_CHUNK_PAIRS="23")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "preyama_sweep.py"), "0", "2"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    last = p.stdout.strip().splitlines()[-1]
    assert last.startswith("merges 600") and last.endswith("bad 0"), p.stdout[-2000:]
