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
