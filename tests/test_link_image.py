"""Link images on the host (include/mz_amd.h: mz_link_pack / mz_link_parts / mz_link_assemble; no GPU): an image expands to the
packed jobs -- classes of the caller's bytes, the caller's band, in all three band formats --, and a result image (here: computed by the
oracle, tests/linkfmt.py) assembles into the reference's merged columns over the caller's OWN bytes."""
import numpy as np
import pytest

import inputs
import linkfmt
from multiz_amd import api
from oracle import mzoracle as mo

CLASS_OF = np.full(256, 5, np.uint8)
for k, ch in enumerate("ACGT"):
    CLASS_OF[ord(ch)] = CLASS_OF[ord(ch.lower())] = k
CLASS_OF[ord("-")] = 4


def _pairs(seed, n):
    rng = np.random.default_rng(seed)
    pairs, tries = [], 0
    while len(pairs) < n:
        tries += 1
        assert tries < 20 * n, "the generator's bands are being refused"
        K, L = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        M, N = int(rng.integers(3, 200)), int(rng.integers(3, 200))
        kind = len(pairs) % 4
        if kind == 3:                                            # full matrix on a lopsided pair: one step of hundreds (raw bounds)
            M, N = int(rng.integers(2, 6)), int(rng.integers(300, 500))
            A, B, _, _ = inputs.make_pair(rng, K, L, M, N, 30, "diag", mo.smooth)
            LB, RB = np.zeros(M + 1, np.int32), np.full(M + 1, N, np.int32)
            LB[1:] = 1
            LB[M] = N - 20
        elif kind == 2:                                          # a wandering band with steps of 16..60 (a byte per step)
            A, B, LB, RB = inputs.make_pair(rng, K, L, M, N, 30, "diag", mo.smooth)
            LB, RB = LB.copy(), RB.copy()
            r = int(rng.integers(1, M))
            RB[r:] = np.minimum(N, RB[r:] + int(rng.integers(16, 60)))
            RB[M] = N
        else:
            A, B, LB, RB = inputs.make_pair(rng, K, L, M, N, 30, "diag" if kind else "wander", mo.smooth)
        if mo.check(M, N, LB, RB)[0] == 0:
            pairs.append((A, B, np.ascontiguousarray(LB, np.int32), np.ascontiguousarray(RB, np.int32)))
    return pairs


def _jobs(pairs):
    jobs = np.zeros(len(pairs), dtype=api.JOB_DT)
    for i, (A, B, LB, RB) in enumerate(pairs):
        jobs[i] = (A.shape[1], B.shape[1], A.shape[0], B.shape[0], A.ctypes.data, B.ctypes.data, LB.ctypes.data, RB.ctypes.data)
    return jobs


def test_image_expands_to_the_jobs():
    pairs = _pairs(5, 40)
    desc, image, exc = api.link_pack(_jobs(pairs))
    assert exc.size > 0 and desc[0] == 40
    b = linkfmt.decode_image(desc, image, exc)
    fmts = set(image[api.link_parts(desc)["fmt"]:][:40].tolist())
    assert fmts == {0, 1, 2}                                     # raw, a byte per step, a nibble per step: all three travelled
    for p, (A, B, LB, RB) in enumerate(pairs):
        M, K = A.shape
        N, L = B.shape
        assert (b["K"][p], b["L"][p], b["M"][p], b["N"][p]) == (K, L, M, N)
        a0, b0, d0 = int(b["offA"][p]), int(b["offB"][p]), int(b["offBand"][p])
        assert np.array_equal(CLASS_OF[b["poolA"][a0: a0 + K * M]], CLASS_OF[A.ravel()])
        assert np.array_equal(CLASS_OF[b["poolB"][b0: b0 + L * N]], CLASS_OF[B.ravel()])
        assert np.array_equal(b["poolLB"][d0: d0 + M + 1], LB) and np.array_equal(b["poolRB"][d0: d0 + M + 1], RB)
    # bytes per pair: well under the pools' (K*M + L*N bytes + 8 bytes per band row)
    pools = sum(A.size + B.size + 8 * (A.shape[0] + 1) for A, B, _, _ in pairs)
    assert image.size + exc.size < 0.75 * pools


def test_result_image_assembles_over_the_callers_bytes():
    pairs = _pairs(6, 30)
    jobs = _jobs(pairs)
    desc, image, exc = api.link_pack(jobs)
    res, cells, failed = linkfmt.oracle_result_image(desc, image, exc)
    assert failed == 0 and cells == sum(mo.band_cells(p[2], p[3]) for p in pairs)
    outs = api.link_assemble(jobs, res)
    try:
        for p, (A, B, LB, RB) in enumerate(pairs):
            want = mo.yama(A, B, LB, RB)
            W = A.shape[1] + B.shape[1]
            assert outs["status"][p] == 0 and outs["OM"][p] == want.OM and tuple(outs["score"][p]) == tuple(want.final)
            got = np.ctypeslib.as_array((api.C.c_uint8 * (want.OM * W)).from_address(int(outs["cols"][p])))
            assert np.array_equal(got, want.cols.ravel()), p      # the caller's own bytes (case, odd letters), not the classes' letters
    finally:
        api.free_outs(outs)
    # an image that does not belong to the jobs is refused, not followed
    with pytest.raises(RuntimeError, match="too short|does not belong"):
        api.link_assemble(jobs, res[: 64 + 40 * 30 - 8])
    bad = res.copy()
    bad[64: 64 + 40 * 30].view(api.RES_DT)["off"][3] = 1 << 40
    with pytest.raises(RuntimeError, match="does not belong"):
        api.link_assemble(jobs, bad)


def test_empty_and_refused_jobs():
    desc, image, exc = api.link_pack(np.zeros(0, dtype=api.JOB_DT))
    assert desc[0] == 0 and exc.size == 0
    pairs = _pairs(7, 3)
    jobs = _jobs(pairs)
    jobs["LB"][1] = 0                                            # a NULL array: the plan will see M = N = 0 (MZ_E_SHAPE), nothing is read
    desc, image, exc = api.link_pack(jobs)
    b = linkfmt.decode_image(desc, image, exc)
    assert b["M"][1] == 0 and b["N"][1] == 0 and b["M"][0] == pairs[0][0].shape[0]
