#!/usr/bin/env python3
"""Generate tests/golden/yama_golden.npz from the COMPILED REFERENCE (oracle/_ref/libref.so).

Run in the build container only (needs /root/reference to have been compiled by
`make -C oracle ref`).  The output is data: seeded inputs + the reference's own outputs
(OM and merged column bytes).  The reference publishes no golden vectors of its own
(SURVEY.md section 4), so these are the pins for the oracle and for the HIP path.

Each case exercises a distinct guard of reference mz_yama.c:
K=1/L=1, K!=L, M<<N, N<<M, lowercase/N bytes, tie-heavy inputs, full band, tiny radius,
M=1/N=1, bands wider than 64 live rows per anti-diagonal, wandering bands.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import inputs  # noqa: E402
from oracle import mzoracle as mo  # noqa: E402


def main():
    ref = mo.Reference()
    rng = np.random.default_rng(20261002)
    cases = []

    def add(tag, A, B, LB, RB):
        rc, cells, _ = mo.check(A.shape[0], B.shape[0], LB, RB)
        if rc:
            return False
        r = ref.yama(A, B, LB, RB)
        cases.append(dict(tag=tag, A=A, B=B, LB=LB.astype(np.int32), RB=RB.astype(np.int32), OM=r.OM, cols=r.cols))
        return True

    def gen(tag, K, L, M, N, R=30, band="diag", **kw):
        for _ in range(20):
            A, B, LB, RB = inputs.make_pair(rng, K, L, M, N, R, band, mo.smooth, **kw)
            if add(tag, A, B, LB, RB):
                return
        raise RuntimeError("could not build a valid case for " + tag)

    # shapes / row counts
    gen("k1l1_small", 1, 1, 40, 44)
    gen("k1l1_200", 1, 1, 200, 200)
    gen("k2l1", 2, 1, 150, 140)
    gen("k2l2_300", 2, 2, 300, 310)
    gen("k3l5", 3, 5, 120, 100)
    gen("k7l2", 7, 2, 90, 130)
    gen("k10l10", 10, 10, 160, 170)
    gen("k1l12", 1, 12, 80, 75)
    # extreme aspect ratios: band clipped by MIN(M,radius)
    gen("m_ll_n", 2, 2, 12, 240)
    gen("n_ll_m", 2, 2, 240, 12)
    gen("m1", 2, 3, 1, 25)
    gen("n1", 3, 2, 25, 1)
    gen("m1n1", 1, 1, 1, 1)
    gen("m2n2", 2, 2, 2, 2)
    gen("tiny_n5", 2, 2, 30, 5, R=5)
    # radius sweep
    gen("r12", 2, 2, 120, 120, R=12)
    gen("r10", 2, 3, 150, 160, R=10)
    gen("r50", 2, 2, 200, 190, R=50)
    gen("r100_wide", 2, 2, 260, 250, R=100)      # > 64 live rows per anti-diagonal
    gen("full_band", 2, 2, 90, 100, band="full")
    gen("full_band_k4", 4, 3, 70, 140, band="full")
    # wandering bands
    for i in range(6):
        gen(f"wander{i}", int(rng.integers(1, 5)), int(rng.integers(1, 5)), int(rng.integers(60, 260)),
            int(rng.integers(60, 260)), R=int(rng.choice([10, 30])), band="wander")
    # byte classes: lowercase / N / other
    gen("odd_bytes", 3, 3, 130, 120, odd=0.5)
    gen("all_odd", 2, 2, 80, 80, odd=1.0)
    gen("dashy", 4, 4, 150, 150, dash=0.45)
    gen("no_dash", 3, 2, 140, 150, dash=0.0, odd=0.0)
    # tie-heavy: homopolymer blocks (every path through a run scores the same)
    for tag, K, L, M, N in (("ties_a", 1, 1, 70, 64), ("ties_b", 2, 2, 90, 100), ("ties_c", 3, 1, 50, 80)):
        A = np.full((M, K), ord("A"), dtype=np.uint8)
        B = np.full((N, L), ord("A"), dtype=np.uint8)
        LB, RB = mo.smooth(*inputs.diag_band(M, N), M, N, 30)
        assert add(tag, A, B, LB, RB)
    # periodic sequence: many co-optimal alignments
    M, N = 120, 110
    A = np.frombuffer((b"ACGT" * 60)[: M * 2], dtype=np.uint8).reshape(M, 2).copy()
    B = np.frombuffer((b"CGTA" * 60)[: N * 2], dtype=np.uint8).reshape(N, 2).copy()
    LB, RB = mo.smooth(*inputs.diag_band(M, N), M, N, 30)
    assert add("periodic", A, B, LB, RB)
    # a few fully random ones at the default radius, ~config-2 row counts, smaller columns
    for i in range(8):
        gen(f"rand{i}", int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(180, 420)),
            int(rng.integers(180, 420)))
    # one config-2 sized and one config-3-rows case
    gen("c2_shape", 2, 2, 1000, 1040)
    gen("c3_rows", 10, 10, 400, 380)

    out = {}
    for i, c in enumerate(cases):
        for k in ("A", "B", "LB", "RB", "cols"):
            out[f"c{i}_{k}"] = c[k]
        out[f"c{i}_OM"] = np.int32(c["OM"])
    out["tags"] = np.array([c["tag"] for c in cases])
    path = os.path.join(HERE, "yama_golden.npz")
    np.savez_compressed(path, **out)
    print(f"{len(cases)} cases -> {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
