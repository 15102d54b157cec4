"""The CPU oracle against the committed golden vectors (outputs of the compiled reference,
tests/golden/make_golden.py).  Runs anywhere, no GPU, no /root/reference."""
import numpy as np
import pytest

from oracle import mzoracle as mo


@pytest.mark.parametrize("variant", ["faithful", "profile"])
def test_oracle_matches_golden(golden, variant):
    assert len(golden) >= 40
    for c in golden:
        r = mo.yama(c["A"], c["B"], c["LB"], c["RB"], variant=variant)
        assert r.rc == 0, c["tag"]
        assert r.OM == c["OM"], c["tag"]
        assert np.array_equal(r.cols, c["cols"]), c["tag"]


def test_faithful_and_profile_agree_on_traceback_bytes(golden):
    # stronger than equal output: every traceback byte and the final (C,D,I) triple agree
    for c in golden:
        a = mo.yama(c["A"], c["B"], c["LB"], c["RB"], variant="faithful", want_tb=True)
        b = mo.yama(c["A"], c["B"], c["LB"], c["RB"], variant="profile", want_tb=True)
        assert np.array_equal(a.tb, b.tb), c["tag"]
        assert np.array_equal(a.final, b.final), c["tag"]


def test_output_properties(golden):
    # OM in [max(M,N), M+N]; removing all-dash columns from the
    # top K rows reproduces A, from the bottom L rows reproduces B
    for c in golden:
        A, B = c["A"], c["B"]
        (M, K), (N, L) = A.shape, B.shape
        r = mo.yama(A, B, c["LB"], c["RB"])
        assert max(M, N) <= r.OM <= M + N
        top, bot = r.cols[:, :K], r.cols[:, K:]
        keep_a = ~(top == ord("-")).all(axis=1)
        keep_b = ~(bot == ord("-")).all(axis=1)
        # inserted columns are all-dash on the A side by construction; A itself has no all-dash column here
        if not (A == ord("-")).all(axis=1).any():
            assert np.array_equal(top[keep_a], A), c["tag"]
        if not (B == ord("-")).all(axis=1).any():
            assert np.array_equal(bot[keep_b], B), c["tag"]


def test_validity_errors():
    # reference mz_yama.c:58-71: each violated precondition is reported, not silently accepted
    M, N = 20, 20
    LB = np.zeros(M + 1, dtype=np.int32)
    RB = np.full(M + 1, N, dtype=np.int32)
    assert mo.check(M, N, LB, RB)[0] == 0
    bad = LB.copy(); bad[0] = 1
    assert mo.check(M, N, bad, RB)[0] == 1
    bad = RB.copy(); bad[M] = N - 1
    assert mo.check(M, N, LB, bad)[0] == 1
    bad = RB.copy(); bad[3] = 5
    assert mo.check(M, N, LB, bad)[0] in (2, 4)
    bad = LB.copy(); bad[5] = 4; bad[6] = 3
    assert mo.check(M, N, bad, RB)[0] == 3
    bad = RB.copy(); bad[5] = 15; bad[6] = 14
    assert mo.check(M, N, LB, bad)[0] == 4
    A = np.full((M, 1), ord("A"), dtype=np.uint8)
    assert mo.yama(A, A, bad, bad).rc != 0


def test_smooth_properties():
    rng = np.random.default_rng(5)
    import inputs
    for _ in range(50):
        M, N, R = int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.integers(0, 50))
        LB, RB = inputs.wander_band(rng, M, N)
        l, r = mo.smooth(LB, RB, M, N, R)
        assert l[0] == 0 and r[M] == N
        assert (np.diff(l) >= 0).all() and (np.diff(r) >= 0).all()
        assert (l >= 0).all() and (r <= N).all()
