#!/usr/bin/env python3
"""tests/golden/below_sentinel.npz: ONE block pair whose scores fall below the reference's "unreachable" sentinel MININT = INT_MIN / 2
(mz_yama.c:29), with what the COMPILED REFERENCE (oracle/_ref/libref.so) makes of it -- OM and a hash of the merged columns.

How the pair was found: `STRIP_STRESS_ROWS=120 python tests/tools/strip_stress.py 1500 <seed> /tmp/x.npz` on a GPU box (blocks of up to 120 rows on
BOTH sides, wide bands, mostly mismatching columns: inputs.random_wide_pair) saves the pairs the product reports as MZ_E_SENTINEL in
/tmp/x_below.npz, smallest first.  This script (build container: needs oracle/_ref) takes the first of them and adds the reference's result.
The product does NOT reproduce that result -- down there the reference's own walk steps outside the band and reads neighbouring rows' traceback
bytes (include/mz_amd.h, MZ_E_SENTINEL) -- it reports the pair with a status of its own; the fixture pins exactly that."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import mzoracle as mo  # noqa: E402

z = np.load(sys.argv[1])
# Of the saved pairs, the first on which the reference HAS a result (on the others its own walk ends in "Error generating edit script.",
# mz_yama.c:275,290 -- the oracle, pinned to it, returns that as a status instead of leaving the process).
for j in range(12):
    if f"A{j}" not in z:
        raise SystemExit("none of the saved pairs has a reference result: run the stress tool with another seed")
    A, B, LB, RB = z[f"A{j}"], z[f"B{j}"], z[f"LB{j}"].astype(np.int32), z[f"RB{j}"].astype(np.int32)
    w = mo.yama(A, B, LB, RB, variant="profile")
    if w.rc == 0:
        break
r = mo.Reference().yama(A, B, LB, RB)
assert r.OM == w.OM and np.array_equal(r.cols, w.cols), "the oracle and the compiled reference disagree on this pair"
h = mo.fnv1a_np(r.cols, mo.fnv1a_np(np.array([r.OM], dtype=np.int32).view(np.uint8)))
np.savez_compressed(os.path.join(HERE, "below_sentinel.npz"), A=A, B=B, LB=LB, RB=RB, OM=np.int32(r.OM), hash=np.uint64(h), final=np.array(w.final, dtype=np.int64))
print("K, L, M, N =", A.shape[1], B.shape[1], A.shape[0], B.shape[0], "OM", r.OM, "final", w.final, "pair", j)
