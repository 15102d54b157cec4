"""Pins the oracle to the REAL reference (oracle/_ref/libref.so, compiled from /root/reference by
oracle/Makefile).  Skipped where that build is absent; the committed golden vectors cover that case."""
import numpy as np
import pytest

import inputs
from oracle import mzoracle as mo

pytestmark = pytest.mark.skipif(not mo.have_reference(), reason="oracle/_ref/libref.so not built")


@pytest.fixture(scope="module")
def ref():
    return mo.Reference()


def test_yama_random_instances(ref):
    rng = np.random.default_rng(77)
    done = 0
    for it in range(500):
        K, L = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        M, N = int(rng.integers(1, 140)), int(rng.integers(1, 140))
        R = int(rng.choice([10, 12, 30, 50]))
        band = str(rng.choice(["diag", "wander", "full"]))
        A, B, LB, RB = inputs.make_pair(rng, K, L, M, N, R, band, mo.smooth,
                                        dash=float(rng.choice([0.0, 0.08, 0.3])), odd=float(rng.choice([0.0, 0.05, 0.6])))
        if mo.check(M, N, LB, RB)[0]:
            continue
        want = ref.yama(A, B, LB, RB)
        for variant in ("faithful", "profile"):
            got = mo.yama(A, B, LB, RB, variant=variant)
            assert got.rc == 0 and got.OM == want.OM and np.array_equal(got.cols, want.cols), (it, variant, K, L, M, N, R, band)
        done += 1
    assert done > 250


def test_smooth_matches_reference(ref):
    rng = np.random.default_rng(3)
    for _ in range(300):
        M, N, R = int(rng.integers(1, 250)), int(rng.integers(1, 250)), int(rng.integers(0, 45))
        LB, RB = inputs.wander_band(rng, M, N)
        a, b = mo.smooth(LB, RB, M, N, R), ref.smooth(LB, RB, M, N, R)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_hoxd85_tables(ref):
    # both score sets: the oracle's tables equal the reference's (mz_scores.c:94-122)
    import ctypes as C
    for which, sc in ((70, mo.scores70()), (85, mo.scores85())):
        ref.scores(which)
        ss = C.POINTER(C.POINTER(C.c_int)).in_dll(ref.lib, "ss")
        gop = C.POINTER(C.c_int).in_dll(ref.lib, "gop")
        for a in range(128):
            row = ss[a]
            for b in range(128):
                assert row[b] == sc.ss[a][b]
        assert [gop[i] for i in range(16)] == list(sc.gop)
        assert C.c_int.in_dll(ref.lib, "gap_extend").value == sc.gap_extend
        assert C.c_int.in_dll(ref.lib, "gap_open").value == sc.gap_open
    ref.scores(70)
