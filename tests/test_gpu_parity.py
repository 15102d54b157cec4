"""Parity tests proper: the HIP path, called through the C ABI (libmzamd.so), against the CPU oracle
and the committed golden vectors.  Integer/byte work: the bar is bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import inputs
from oracle import mzoracle as mo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mz():
    import multiz_amd as m
    m.api.init(0)
    yield m
    m.lib().mz_enable_fast(1)
    m.lib().mz_enable_row(1)


def _kernels(mz, which):
    """2: everything the plan may pick (row-parallel kernel included); 1: wavefront kernels only (tagged /
    fast / exact); 0: exact kernels only"""
    mz.lib().mz_enable_fast(1 if which else 0)
    mz.lib().mz_enable_row(1 if which == 2 else 0)


@pytest.mark.parametrize("fast", [2, 1, 0])
def test_golden_vectors(mz, golden, fast):
    _kernels(mz, fast)
    pairs = [(c["A"], c["B"], c["LB"], c["RB"]) for c in golden]
    res = mz.yama_batch(pairs)
    for c, r in zip(golden, res):
        assert r.status == 0, c["tag"]
        assert r.OM == c["OM"], c["tag"]
        assert np.array_equal(r.cols, c["cols"]), c["tag"]
    _check(res, pairs, [c["tag"] for c in golden], exact_scores=(fast == 0))


def _random_pairs(seed, n, kmax=8, mmax=420):
    rng = np.random.default_rng(seed)
    pairs = []
    while len(pairs) < n:
        K, L = int(rng.integers(1, kmax + 1)), int(rng.integers(1, kmax + 1))
        M, N = int(rng.integers(1, mmax)), int(rng.integers(1, mmax))
        R = int(rng.choice([10, 12, 30, 50, 100]))
        band = str(rng.choice(["diag", "diag", "wander", "full"]))
        A, B, LB, RB = inputs.make_pair(rng, K, L, M, N, R, band, mo.smooth,
                                        dash=float(rng.choice([0.0, 0.08, 0.35])), odd=float(rng.choice([0.0, 0.05, 0.6])))
        if mo.check(M, N, LB, RB)[0] == 0 and (band != "full" or M * N < 40000):
            pairs.append((A, B, LB, RB))
    return pairs


@pytest.mark.parametrize("fast", [2, 1, 0])
def test_random_pairs_one_batch(mz, fast):
    _kernels(mz, fast)
    pairs = _random_pairs(2026 + fast, 400)
    _check(mz.yama_batch(pairs), pairs, exact_scores=(fast == 0))


def test_many_rows(mz):
    # deep sum-of-pairs: 10+10 and lopsided row counts up to the int8 counter limit
    rng = np.random.default_rng(9)
    pairs = []
    for K, L, M, N in ((10, 10, 300, 310), (29, 1, 150, 160), (1, 29, 150, 140), (40, 37, 90, 100), (127, 3, 60, 64), (3, 127, 64, 60)):
        pairs.append(inputs.make_pair(rng, K, L, M, N, 30, "diag", mo.smooth))
    for fast in (1, 0):
        mz.lib().mz_enable_fast(fast)
        _check(mz.yama_batch(pairs), pairs, exact_scores=(fast == 0))


def test_hoxd85_tables(mz):
    # the caller switches tables with init_scores85() (mz_scores.c:109-122); gap_open 600 = 24*25
    rng = np.random.default_rng(85)
    pairs = _random_pairs(85, 60, kmax=4, mmax=200)
    try:
        mz.set_scores_hoxd85()
        res = mz.yama_batch(pairs)
        sc = mo.scores85()
        for (A, B, LB, RB), r in zip(pairs, res):
            want = mo.yama(A, B, LB, RB, sc=sc)
            assert r.status == 0 and r.OM == want.OM and np.array_equal(r.cols, want.cols)
    finally:
        mz.set_scores_hoxd70()
    del rng


def test_invalid_bands_are_reported(mz):
    # reference mz_yama.c:58-71: each violated precondition -> its own status (the yama() wrapper turns
    # these into the reference's message + exit(1)); valid neighbours in the same batch still run
    rng = np.random.default_rng(4)
    good = inputs.make_pair(rng, 2, 2, 60, 64, 30, "diag", mo.smooth)
    A, B, LB, RB = good
    cases = []
    l = LB.copy(); l[0] = 1; cases.append((A, B, l, RB))
    r = RB.copy(); r[-1] -= 1; cases.append((A, B, LB, r))
    r = RB.copy(); r[20] = LB[20] + 3; cases.append((A, B, LB, r))
    l = LB.copy(); l[40] = l[39] - 1 if l[39] > 0 else 0; l[41:] = np.maximum(l[41:], 0); l[39] = l[40] + 2; cases.append((A, B, l, RB))
    r = RB.copy(); r[30] = r[29] - 1; cases.append((A, B, LB, r))
    res = mz.yama_batch([good] + cases + [good])
    assert res[0].status == 0 and res[-1].status == 0 and np.array_equal(res[0].cols, res[-1].cols)
    for (a, b, l, r), got in zip(cases, res[1:-1]):
        rc, _, bad = mo.check(a.shape[0], b.shape[0], l, r)
        assert rc != 0 and got.status == rc, (rc, got.status)
        if rc == 2:
            assert got.badrow == bad
    # row-count limits of this build are reported, not mis-computed
    big = inputs.make_pair(rng, 256, 2, 20, 20, 30, "diag", mo.smooth)
    assert mz.yama_batch([big])[0].status == 16


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4, 5])
def test_exported_yama_dies_like_the_reference(case):
    # the exported yama() itself (mz_yama.h:22) on each invalid band of mz_yama.c:58-71, in a child process: the message on
    # stderr (util.c:21-30: "<argv0 basename>: ..." through fatalf) and the exit code, against the compiled reference's own yama()
    # on the same arrays (oracle/_ref/libref.so) -- byte for byte
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = os.path.join(root, "oracle", "_ref", "libref.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/libref.so not built")
    child = os.path.join(root, "tests", "tools", "yama_fatal_child.py")
    ours = subprocess.run([sys.executable, child, os.path.join(root, "multiz_amd", "libmzamd.so"), str(case)], capture_output=True, text=True, timeout=300)
    want = subprocess.run([sys.executable, child, ref, str(case)], capture_output=True, text=True, timeout=300)
    assert want.returncode == 1 and want.stderr.startswith("multiz: "), (want.returncode, want.stderr)
    assert ours.returncode == want.returncode
    assert [l for l in ours.stderr.splitlines() if l.startswith("multiz: ")] == [l for l in want.stderr.splitlines() if l.startswith("multiz: ")], (ours.stderr, want.stderr)
    assert ours.stdout == want.stdout == ""


def test_yama_dropin_signature(mz):
    # the reference-signature entry point itself: 1-based column pointer arrays in, two malloc blocks out
    lib = mz.lib()
    rng = np.random.default_rng(12)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for _ in range(5):
        A, B, LB, RB = inputs.make_pair(rng, 3, 2, int(rng.integers(30, 200)), int(rng.integers(30, 200)), 30, "diag", mo.smooth)
        (M, K), (N, L) = A.shape, B.shape
        pa, pb = (C.c_void_p * (M + 1))(), (C.c_void_p * (N + 1))()
        for i in range(1, M + 1):
            pa[i] = A.ctypes.data + (i - 1) * K
        for i in range(1, N + 1):
            pb[i] = B.ctypes.data + (i - 1) * L
        oal, om = C.POINTER(C.c_void_p)(), C.c_int(0)
        lb, rb = LB.astype(np.int32), RB.astype(np.int32)
        lib.yama(pa, C.c_int(K), C.c_int(M), pb, C.c_int(L), C.c_int(N), C.c_void_p(lb.ctypes.data),
                 C.c_void_p(rb.ctypes.data), C.byref(oal), C.byref(om))
        want = mo.yama(A, B, LB, RB)
        got = np.frombuffer(C.string_at(oal[1], om.value * (K + L)), dtype=np.uint8).reshape(om.value, K + L)
        assert om.value == want.OM and np.array_equal(got, want.cols)
        libc.free(C.c_void_p(oal[1]))                                         # free(OAL[1]); free(OAL+1);
        libc.free(C.c_void_p(C.addressof(oal.contents) + C.sizeof(C.c_void_p)))


def _hash(cols, om):
    return mo.fnv1a_np(cols, mo.fnv1a_np(np.array([om], dtype=np.int32).view(np.uint8)))


@pytest.mark.parametrize("cfg,pairs", [("c2", 10000), ("c3", 1000)])
def test_config_sized_batches_device_resident(mz, cfg, pairs):
    # BASELINE.json shapes through the device-resident API (what bench.py times): every pair against the
    # oracle by hash of (OM, merged columns), plus the size-independent properties of the merge
    from multiz_amd import synth
    c = synth.CONFIGS[cfg]
    batch = synth.make_batch(pairs, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=7)
    _kernels(mz, 2)
    db = mz.DevBatch(batch)
    db.run()
    res = db.results()
    assert (res["status"] == 0).all()
    om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=min(64, os.cpu_count() or 8))
    assert bad == 0 and cells == int(res["cells"].sum())
    host_out = db.out.cpu().numpy()
    for i in range(pairs):
        K, L, M, N = (int(batch[k][i]) for k in ("K", "L", "M", "N"))
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        assert m_ == om[i] and max(M, N) <= m_ <= M + N
        g = host_out[o0: o0 + m_ * (K + L)]
        assert _hash(g, m_) == int(hs[i]), i
    # round trip on a few: dropping the inserted (all-dash) half-columns gives back A and B
    for i in range(0, pairs, max(1, pairs // 20)):
        A, B, _, _ = synth.pair_of(batch, i)
        K, L = A.shape[1], B.shape[1]
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        g = host_out[o0: o0 + m_ * (K + L)].reshape(m_, K + L)
        top, bot = g[:, :K], g[:, K:]
        assert np.array_equal(top[~(top == 45).all(axis=1)], A)
        assert np.array_equal(bot[~(bot == 45).all(axis=1)], B)


def test_config4_tree_workload(mz):
    # BASELINE.json configs[3]: block pairs of a 30-leaf caterpillar+balanced guide tree (K = p, L = q at a node with
    # p, q leaves below its children; M,N~U[200,1000]) -- 20 000 pairs of one GPU's share, every pair against the
    # oracle by hash of (OM, merged columns), through the device-resident API that bench.py --config c4 times
    from multiz_amd import synth
    n = 20000
    c = synth.CONFIGS["c4"]
    batch = synth.make_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=3 * 125000)
    tk, tl = synth.tree_nodes()
    assert set(zip(batch["K"].tolist(), batch["L"].tolist())) == set(zip(tk.tolist(), tl.tolist()))   # all 29 node shapes
    assert batch["K"].max() == 29 and (batch["K"] + batch["L"]).max() == 30
    _kernels(mz, 2)
    db = mz.DevBatch(batch)
    db.run()
    res = db.results()
    assert (res["status"] == 0).all()
    assert set(res["mode"].tolist()) <= {5, 6, 7, 8}, np.bincount(res["mode"])        # all on the row-parallel kernels
    om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=min(64, os.cpu_count() or 8))
    assert bad == 0 and cells == int(res["cells"].sum())
    host_out = db.out.cpu().numpy()
    W = batch["K"].astype(np.int64) + batch["L"]
    mism = 0
    for i in range(n):
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        mism += m_ != om[i] or _hash(host_out[o0: o0 + m_ * int(W[i])], m_) != int(hs[i])
    assert mism == 0, mism
    # the literal O(K*L)-per-cell restatement on a few pairs of the deepest nodes
    deep = np.flatnonzero(batch["K"] * batch["L"] >= 29)[:6]
    for i in deep:
        A, B, LB, RB = synth.pair_of(batch, int(i))
        w = mo.yama(A, B, LB, RB)
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        assert m_ == w.OM and np.array_equal(host_out[o0: o0 + m_ * int(W[i])].reshape(m_, -1), w.cols)


def test_every_kernel_family_is_exercised(mz):
    # mixed shapes through the device-resident API: the plan must send narrow bands to the row-parallel kernel
    # (mode 5), wide-but-low ones to its transposed form (mode 6), the rest to the wavefront kernels, and every
    # pair must match the oracle whatever kernel took it.  Band widths sit on the eligibility edges (62 / 63).
    from multiz_amd import synth
    rng = np.random.default_rng(77)
    pairs = []
    for K, L, M, N, R, band in ((2, 2, 300, 300, 30, "diag"), (3, 1, 200, 260, 30, "diag"), (1, 4, 260, 200, 30, "diag"),
                                (2, 3, 70, 64, 31, "diag"), (2, 2, 64, 70, 31, "diag"), (4, 4, 150, 150, 12, "wander"),
                                (2, 2, 150, 230, 25, "wander"), (2, 2, 230, 150, 25, "wander"), (5, 2, 90, 400, 30, "diag"),
                                (2, 5, 400, 90, 30, "diag"), (2, 2, 500, 500, 60, "diag"), (1, 1, 63, 63, 31, "diag"),
                                (2, 2, 40, 45, 10, "diag"), (6, 6, 129, 128, 30, "diag"),
                                (20, 16, 300, 280, 30, "diag"), (16, 20, 280, 300, 30, "diag")):   # rotate-scan modes 7 / 8
        for _ in range(6):
            A, B, LB, RB = inputs.make_pair(rng, K, L, M, N, R, band, mo.smooth,
                                            dash=float(rng.choice([0.0, 0.08, 0.3])), odd=float(rng.choice([0.0, 0.05])))
            if mo.check(M, N, LB, RB)[0] == 0:
                pairs.append((A, B, LB, RB))
    batch = synth.pack_pairs(pairs)
    _kernels(mz, 2)
    db = mz.DevBatch(batch)
    db.run()
    res = db.results()
    assert (res["status"] == 0).all()
    hist = np.bincount(res["mode"], minlength=9)
    assert hist[5] > 0 and hist[6] > 0 and hist[7] + hist[8] > 0 and hist[:5].sum() > 0, hist
    host_out = db.out.cpu().numpy()
    for i, (A, B, LB, RB) in enumerate(pairs):
        want = mo.yama(A, B, LB, RB)
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        assert m_ == want.OM, (i, int(res["mode"][i]))
        got = host_out[o0: o0 + m_ * (A.shape[1] + B.shape[1])].reshape(m_, -1)
        assert np.array_equal(got, want.cols), (i, int(res["mode"][i]))


def test_plan_lists_hold_every_valid_pair_once_largest_first(mz):
    # The DP kernels take their pairs from the plan's lists (wavefront / 128+ rows / lagged / row-parallel, one after the other;
    # kernels/plan.inc: a counting sort on the half-octave class of the band cells, largest class first).  A pair missing
    # from its list would never be computed, one listed twice computed twice: the lists together must be a permutation of
    # the valid pairs, each list must hold exactly the pairs of its kind, and its classes must not grow along the list.
    from multiz_amd import synth
    rng = np.random.default_rng(5)
    pairs = []
    for K, L, M, N, R, band in ((2, 2, 300, 300, 30, "diag"), (2, 2, 900, 1000, 30, "diag"), (3, 1, 120, 150, 30, "diag"),
                                (1, 4, 260, 200, 30, "diag"), (2, 2, 150, 230, 25, "wander"), (2, 2, 500, 500, 60, "diag"),
                                (2, 2, 40, 45, 10, "diag"), (20, 16, 300, 280, 30, "diag"), (2, 2, 2000, 2100, 30, "diag")):
        for _ in range(40):
            A, B, LB, RB = inputs.make_pair(rng, K, L, M + int(rng.integers(0, 40)), N + int(rng.integers(0, 40)), R, band, mo.smooth)
            if rng.random() < 0.05:
                LB = LB.copy(); LB[0] = 1                 # an invalid band: the pair fails in the plan and is on no list
            pairs.append((A, B, LB, RB))
    order = rng.permutation(len(pairs))
    batch = synth.pack_pairs([pairs[i] for i in order])
    for lag_pairs in (False, True):
        if lag_pairs:                                      # the bench's indel mix: mostly lagged pairs
            c = synth.CONFIGS["c2i"]
            batch = synth.make_batch(3000, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], indel=c["indel"])
        _kernels(mz, 2)
        db = mz.DevBatch(batch)
        db.run()
        res = db.results()
        n = len(res["status"])
        tot = db._view(db.c.totals, 16, np.int64)
        plist = db._view(db.c.packList, n, np.int32)
        ok = res["status"] == 0
        nwf, nwide, nlag = int(tot[5] & 0xffffffff), int(tot[8] & 0xffffffff), int(tot[8] >> 32)
        nvalid = int(ok.sum())
        assert n - nvalid == int(tot[3])
        listed = plist[:nvalid]
        assert np.array_equal(np.sort(listed), np.flatnonzero(ok)), "the lists are not a permutation of the valid pairs"
        mode = res["mode"]
        kind = np.where(mode < 5, 0, np.where((mode == 9) | (mode == 10), 1, np.where(mode == 11, 2, 3)))
        bounds = [0, nwf, nwf + nwide, nwf + nwide + nlag, nvalid]
        cells = np.maximum(res["cells"], 256)
        lg = np.floor(np.log2(cells)).astype(np.int64)
        cls = np.minimum(2 * (lg - 8) + ((cells >> (lg - 1)) & 1), 31)
        for k in range(4):
            part = listed[bounds[k]: bounds[k + 1]]
            assert (kind[part] == k).all(), k
            assert (np.diff(cls[part]) <= 0).all(), f"list {k} is not ordered by size class"
        if lag_pairs:
            assert nlag > 2000 and bounds[4] - bounds[3] > 0


def test_large_score_magnitudes(mz):
    # columns whose dash pattern flips from one column to the next make every row pair open a gap at every step:
    # final scores of -1e7 .. -9e7, within a small factor of the magnitude bounds under which the plan admits
    # the tagged / row-parallel (ring-lift and rotate-scan) kernels.  Every kernel set must still match the oracle.
    from multiz_amd import synth

    def flip_pair(K, L, M, N, R, pat):
        r, i = np.arange(M)[:, None], np.arange(K)[None, :]
        A = np.where(((r + i) % 2 == 0) if pat == 0 else ((r // 2 + i) % 2 == 0), ord("A"), ord("-")).astype(np.uint8)
        c, j = np.arange(N)[:, None], np.arange(L)[None, :]
        B = np.where((c + j) % 2 == 1, ord("C"), ord("-")).astype(np.uint8)
        for X in (A, B):
            X[(X == 45).all(axis=1), 0] = ord("G")
        LB, RB = mo.smooth(*inputs.diag_band(M, N), M, N, R)
        return A, B, LB, RB

    pairs = [flip_pair(K, L, M, N, 30, pat)
             for (K, L, M, N) in ((8, 8, 740, 740), (8, 8, 700, 760), (16, 16, 700, 700), (20, 20, 900, 860),
                                  (30, 30, 420, 400), (4, 4, 2900, 2900), (12, 10, 1500, 1400))
             for pat in (0, 1)]
    want = [mo.yama(*p) for p in pairs]
    assert min(int(w.final.min()) for w in want) < -8e7
    batch = synth.pack_pairs(pairs)
    seen = np.zeros(9, dtype=np.int64)
    for which in (2, 1, 0):
        _kernels(mz, which)
        db = mz.DevBatch(batch)
        db.run()
        res = db.results()
        assert (res["status"] == 0).all()
        seen += np.bincount(res["mode"], minlength=9)
        out = db.out.cpu().numpy()
        for i, (p, w) in enumerate(zip(pairs, want)):
            m_, o0 = int(res["om"][i]), int(res["offOut"][i])
            assert m_ == w.OM, (which, i, int(res["mode"][i]))
            assert np.array_equal(out[o0: o0 + m_ * (p[0].shape[1] + p[1].shape[1])].reshape(m_, -1), w.cols), (which, i, int(res["mode"][i]))
    assert seen[5] and seen[6] and seen[7] and seen[2] + seen[3] and seen[0], seen


def test_pipelined_batches_rotating_workspaces(mz):
    # mz_dev_run_async(): walk + emit of batch k overlap plan + DP of batch k+1 on a second stream; two
    # alternating workspaces.  After mz_dev_wait() both must hold exactly what the serial form produces.
    from multiz_amd import synth
    c = synth.CONFIGS["c2"]
    batch = synth.make_batch(2000, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=123)
    _kernels(mz, 2)
    ref = mz.DevBatch(batch)
    ref.run()
    r0 = ref.results()
    want = ref.out.cpu().numpy().copy()
    a = mz.DevBatch(batch)
    b = a.alternate()
    c3 = a.alternate()
    for k in range(7):                                   # two alternating workspaces ...
        (a if k % 2 == 0 else b).run_async()
    for k in range(7):                                   # ... then three rotating ones
        (a, b, c3)[k % 3].run_async()
    a.wait()
    width = batch["K"].astype(np.int64) + batch["L"]
    for w in (a, b, c3):
        r = w.results()
        assert np.array_equal(r["status"], r0["status"]) and np.array_equal(r["om"], r0["om"])
        assert np.array_equal(r["final3"], r0["final3"]) and np.array_equal(r["offOut"], r0["offOut"])
        got = w.out.cpu().numpy()
        for i in range(len(width)):                      # (the 16-byte padding between pairs is never written)
            o0, nb = int(r0["offOut"][i]), int(r0["om"][i]) * int(width[i])
            assert np.array_equal(got[o0: o0 + nb], want[o0: o0 + nb]), i


def test_long_block_regime(mz):
    # configs[4] shape (R=30, ~100k x 100k columns): traceback spills to HBM (6 MB per pair); 64 pairs of the batch
    from multiz_amd import synth
    npairs = 64
    batch = synth.make_batch(npairs, 2, 2, 95000, 105000, 30, first_pair=3)
    _kernels(mz, 2)
    db = mz.DevBatch(batch)
    db.run()
    res = db.results()
    assert (res["status"] == 0).all()
    om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=min(64, os.cpu_count() or 8))
    assert bad == 0 and cells == int(res["cells"].sum())
    host_out = db.out.cpu().numpy()
    for i in range(npairs):
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        assert m_ == om[i] and _hash(host_out[o0: o0 + m_ * 4], m_) == int(hs[i])
    # these pairs run on the row-parallel kernels, whose scores are re-based every 64 rows: the final triple
    # (re-assembled from the running 64-bit offset) must still be the reference's, about -1e7 here
    assert set(res["mode"].tolist()) <= {5, 6}
    for i in range(3):
        want = mo.yama(*synth.pair_of(batch, i), variant="profile")
        assert res["final3"][i].max() == want.final.max()
        live = want.final > -(1 << 29)
        assert np.array_equal(res["final3"][i][live], want.final[live])


def test_no_global_score_range_limit(mz):
    # The reference has no bound on K*L*(M+N) (mz_yama.c:50-71).  Round 1 refused every pair whose WORST-CASE path
    # score could pass 2^30; now only the guard-dropping kernels are restricted (by the plan), the row-parallel
    # kernels re-base and the exact ones compute in the reference's int32.  Shapes that used to die with MZ_E_RANGE:
    # a 40-way merge of 13 000 columns, a 100-way block pair (K*L = 2 500) of 4 000 columns, a long 20-way pair on
    # the exact kernels.  Real scores stay far inside int32 (checked against the oracle's finals).
    from multiz_amd import synth
    rng = np.random.default_rng(31)
    pairs = [inputs.make_pair(rng, 20, 20, 6500, 6480, 30, "diag", mo.smooth),
             inputs.make_pair(rng, 50, 50, 2000, 2000, 30, "diag", mo.smooth),
             inputs.make_pair(rng, 29, 1, 9000, 9100, 30, "diag", mo.smooth),
             inputs.make_pair(rng, 10, 10, 15000, 14900, 30, "diag", mo.smooth),
             inputs.make_pair(rng, 60, 40, 700, 720, 12, "wander", mo.smooth)]
    want = [mo.yama(*p, variant="profile") for p in pairs]
    faithful = mo.yama(*pairs[1])                        # O(K*L) per cell: the literal restatement, once
    assert faithful.OM == want[1].OM and np.array_equal(faithful.cols, want[1].cols)
    batch = synth.pack_pairs(pairs)
    seen = np.zeros(16, dtype=np.int64)
    for which in (2, 1, 0):
        _kernels(mz, which)
        db = mz.DevBatch(batch)
        db.run()
        res = db.results()
        assert (res["status"] == 0).all(), (which, res["status"])
        seen += np.bincount(res["mode"], minlength=16)
        out = db.out.cpu().numpy()
        for i, (p, w) in enumerate(zip(pairs, want)):
            m_, o0 = int(res["om"][i]), int(res["offOut"][i])
            assert m_ == w.OM, (which, i, int(res["mode"][i]))
            assert np.array_equal(out[o0: o0 + m_ * (p[0].shape[1] + p[1].shape[1])].reshape(m_, -1), w.cols), (which, i, int(res["mode"][i]))
            assert res["final3"][i].max() == w.final.max(), (which, i)
    assert seen[5] + seen[6] > 0 and seen[7] + seen[8] > 0 and seen[0] > 0, seen
    _kernels(mz, 2)


def test_blocks_of_more_than_127_rows(mz):
    # K, L up to 255: the exact kernels on int16 gap vectors (MZ_MODE_WIDE, and WIDESTRIP for bands the rolling
    # wavefront cannot hold), against the oracle -- the literal O(K*L) restatement on the small ones
    from multiz_amd import synth
    rng = np.random.default_rng(150)
    shapes = [(150, 3, 300, 310, 30, "diag"), (3, 200, 260, 250, 30, "diag"), (140, 130, 120, 124, 30, "diag"),
              (255, 1, 90, 90, 30, "diag"), (1, 255, 80, 95, 12, "wander"), (128, 128, 70, 66, 30, "full"),
              (160, 2, 400, 380, 100, "diag"), (2, 129, 150, 150, 60, "diag")]
    pairs = [inputs.make_pair(rng, K, L, M, N, R, band, mo.smooth, dash=float(rng.choice([0.05, 0.3])), odd=0.05)
             for (K, L, M, N, R, band) in shapes]
    pairs = [p for p in pairs if mo.check(p[0].shape[0], p[1].shape[0], p[2], p[3])[0] == 0]
    batch = synth.pack_pairs(pairs)
    db = mz.DevBatch(batch)
    db.run()
    res = db.results()
    assert (res["status"] == 0).all(), res["status"]
    assert set(res["mode"].tolist()) <= {9, 10} and (res["mode"] == 9).any() and (res["mode"] == 10).any(), res["mode"]
    out = db.out.cpu().numpy()
    for i, p in enumerate(pairs):
        w = mo.yama(*p, variant="faithful" if p[0].shape[0] * p[1].shape[0] < 30000 else "profile")
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        assert m_ == w.OM, i
        assert np.array_equal(out[o0: o0 + m_ * (p[0].shape[1] + p[1].shape[1])].reshape(m_, -1), w.cols), i
        assert np.array_equal(res["final3"][i], w.final), i          # exact kernels: the reference's own triple
    # and through the host path, next to ordinary pairs
    mixed = pairs[:3] + [inputs.make_pair(rng, 2, 2, 200, 210, 30, "diag", mo.smooth)]
    for p, r in zip(mixed, mz.yama_batch(mixed)):
        w = mo.yama(*p, variant="profile")
        assert r.status == 0 and r.OM == w.OM and np.array_equal(r.cols, w.cols)


def _indel_band_pair(rng, rate, mean_len=3.0):
    """C2-shaped pair whose band is what pre_yama derives from blocks with indels against the shared reference row
    (mz_preyama.c:240-258, then smooth): the centre stands still over columns only the first block has and jumps over
    columns only the second block has"""
    M = int(rng.integers(900, 1101))
    centre = np.zeros(M + 1, dtype=np.int64)
    c, i = 0, 1
    while i <= M:
        u = rng.random()
        if u < rate / 2 and i > 1:
            for _ in range(min(int(rng.geometric(1.0 / mean_len)), M - i + 1)):
                centre[i] = c; i += 1
            continue
        if u < rate:
            c += int(rng.geometric(1.0 / mean_len))
        c += 1
        centre[i] = c; i += 1
    N = int(max(c, 11))
    LB = np.minimum(centre, N).astype(np.int32); RB = LB.copy(); LB[0] = 0; RB[M] = N
    LB, RB = mo.smooth(LB, RB, M, N, 30)
    A = inputs.random_block(rng, M, 2, dash=0.08, odd=0.05)
    return A, inputs.noisy_copy(rng, A, N, 2, dash=0.08), LB, RB


@pytest.mark.parametrize("events", [2, 10, 30])
def test_bands_with_indels(mz, events):
    # Real MAF blocks: insertions in either block widen the rows or heighten the columns of the band around them, so
    # pairs leave the plain row-parallel kernels (rows <= 63 wide or columns <= 63 high) as the indel rate grows -- for
    # the lagged row-parallel kernel (MZ_MODE_LAG, kernels/lag.inc), and the tagged wavefront where the band does not
    # fit that either.  Every pair, whatever kernel the plan picks, against the oracle by hash; the mix is reported by
    # tests/tools/indel_bands.py
    from multiz_amd import synth
    rng = np.random.default_rng(100 + events)
    n = 1500
    pairs = [_indel_band_pair(rng, events / 1000.0) for _ in range(n)]
    batch = synth.pack_pairs(pairs)
    _kernels(mz, 2)
    db = mz.DevBatch(batch)
    db.run()
    res = db.results()
    assert (res["status"] == 0).all()
    hist = np.bincount(res["mode"], minlength=13)
    print("modes", hist)
    assert hist[3] + hist[5] + hist[6] + hist[11] == n, hist            # tagged wavefront, ROW, COL, LAG
    if events >= 10:
        assert hist[11] > n // 2 and hist[3] < n // 5, hist             # bands with wide rows run lagged, few are left to the wavefront
    om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=min(64, os.cpu_count() or 8))
    assert bad == 0 and cells == int(res["cells"].sum())
    out = db.out.cpu().numpy()
    for i in range(n):
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        assert m_ == om[i] and _hash(out[o0: o0 + m_ * 4], m_) == int(hs[i]), (i, int(res["mode"][i]))


def _lag_steps(LB, RB):
    """LAG steps per 64-column period from their definition (kernels/lag.inc): 1 + the most rows any column c of the
    period before shares with column c+64, at least 1"""
    M = len(LB) - 1; N = int(RB[M])
    tlo = np.searchsorted(RB, np.arange(N + 1), side="left")            # first row with RB >= c
    thi = np.searchsorted(LB, np.arange(N + 1), side="right") - 1       # last row with LB <= c
    step = np.ones((N >> 6) + 1, dtype=np.int64); step[0] = 0
    for k in range(len(step) - 1):
        c = np.arange(64 * k, min(64 * k + 64, N + 1 - 64))
        if len(c):
            step[k + 1] = max(1, int((thi[c] - tlo[c + 64] + 2).max()))
    return step


def test_lagged_schedule(mz):
    # MZ_MODE_LAG: the plan's table (one LAG per 64-column period, in the pair's prep slice) equals the definition,
    # pairs are taken exactly when the band fits the kernel's limits, and the result equals the oracle's -- on bands
    # with long indels (steps up to the limit), short pairs (one or two periods), radii from 10 to 30, and on random
    # monotone bands (both edges random walks with jumps)
    from multiz_amd import synth
    rng = np.random.default_rng(77)
    pairs = []
    for i in range(900):
        M = int(rng.integers(70, 500))
        if i % 3 == 2:                                                   # a third: bands no aligner would produce
            LB, RB, N = inputs.random_walk_band(rng, M)
        else:
            LB, RB, N = inputs.indel_band(rng, M, rng.choice([0.01, 0.03, 0.06, 0.1]), rng.choice([3.0, 6.0, 10.0]),
                                          int(rng.integers(10, 31)))    # (yama wants rows of at least 11 columns)
        A = inputs.random_block(rng, M, 2, dash=0.1, odd=0.05)
        pairs.append((A, inputs.noisy_copy(rng, A, N, 3, dash=0.1), LB, RB))
    batch = synth.pack_pairs(pairs)
    _kernels(mz, 2)
    db = mz.DevBatch(batch)
    db.run()
    res = db.results()
    prep = db.prep.cpu().numpy()
    assert (res["status"] == 0).all()
    nlag = 0
    for i, (A, B, LB, RB) in enumerate(pairs):
        M, N = len(LB) - 1, int(RB[-1])
        step = _lag_steps(LB, RB)
        wide = int((RB - LB).max()) > 62 and not all(LB[r + 63] > RB[r] for r in range(M + 1 - 63))   # fits neither ROW nor COL
        fits = (step.max() <= 31 and (len(step) < 2 or int((step[1:] + step[:-1]).max()) <= 32) and
                int((RB - LB).max()) <= 126 and RB[0] <= 62)
        if res["mode"][i] == 11:
            nlag += 1
            assert fits, i
            o = int(res["offPrep"][i])
            lam = np.cumsum(step)
            assert np.array_equal(prep[o: o + len(step)], lam), (i, prep[o: o + len(step)], lam)
            assert prep[o + len(step)] == M + lam[-1]
            assert res["edgeHi"][i] == min(int(np.argmax(RB == N)) + lam[N >> 6], M + lam[LB[M] >> 6])
        elif wide and res["mode"][i] not in (5, 6):
            assert not fits, (i, int(res["mode"][i]), step)
        w = mo.yama(A, B, LB, RB, variant="profile")
        m_, o0 = int(res["om"][i]), int(res["offOut"][i])
        assert m_ == w.OM and np.array_equal(db.out[o0: o0 + m_ * 5].cpu().numpy().reshape(m_, 5), w.cols), (i, int(res["mode"][i]))
    assert nlag > 50, nlag
    # the same batch without the caller's hints (mz_dev_batch.dp_hint = dp_grid = walk_hint = 0: every DP kernel launched, one
    # after the other, the device choosing the walk) gives the same bytes
    db2 = mz.DevBatch(batch)
    db2.c.dp_hint = 0
    db2.c.dp_grid = 0
    db2.c.dp_rows = 0
    db2.c.walk_hint = 0
    db2.run()
    res2 = db2.results()
    assert np.array_equal(res2["om"], res["om"]) and np.array_equal(res2["mode"], res["mode"])
    o1, o2 = db.out.cpu().numpy(), db2.out.cpu().numpy()
    for i in range(len(pairs)):                                          # (the slack between the pairs' slices is not written)
        a, e = int(res["offOut"][i]), int(res["offOut"][i]) + 5 * int(res["om"][i])
        assert res2["offOut"][i] == res["offOut"][i] and np.array_equal(o1[a:e], o2[a:e]), i


def test_lagged_kernel_random_blocks(mz):
    # blocks of 1-6 rows, short and long pairs, long indels, many dashes -- among them bands whose last rows are wide (a
    # column that stays in the band down to row M while the column 64 to its right, lagged, still has rows to come:
    # the lane leaves it at the first row beyond M).  Every pair against the oracle by hash.
    from multiz_amd import synth
    rng = np.random.default_rng(3)
    pairs = [inputs.random_indel_pair(rng) for _ in range(4000)]
    for _ in range(1000):                                                # and bands no aligner would produce
        M = int(rng.integers(40, 400))
        LB, RB, N = inputs.random_walk_band(rng, M)
        A = inputs.random_block(rng, M, int(rng.integers(1, 5)), dash=0.1, odd=0.05)
        pairs.append((A, inputs.noisy_copy(rng, A, N, int(rng.integers(1, 5)), dash=0.1), LB, RB))
    batch = synth.pack_pairs(pairs)
    _kernels(mz, 2)
    db = mz.DevBatch(batch)
    db.run()
    res = db.results()
    assert (res["status"] == 0).all()
    hist = np.bincount(res["mode"], minlength=13)
    assert hist[11] > 400, hist
    om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=min(64, os.cpu_count() or 8))
    assert bad == 0 and cells == int(res["cells"].sum())
    out = db.out.cpu().numpy()
    for i, (A, B, _, _) in enumerate(pairs):
        m_, o0, W = int(res["om"][i]), int(res["offOut"][i]), A.shape[1] + B.shape[1]
        assert m_ == om[i] and _hash(out[o0: o0 + m_ * W], m_) == int(hs[i]), (i, int(res["mode"][i]))
    # the same pairs from host buffers (mz_yama_batch: chunks in flight, DP kernels of a chunk side by side)
    lagged = [i for i in range(len(pairs)) if res["mode"][i] == 11][:300]
    for i, r in zip(lagged, mz.yama_batch([pairs[i] for i in lagged])):
        W = pairs[i][0].shape[1] + pairs[i][1].shape[1]
        assert r.status == 0 and r.OM == om[i] and _hash(np.ascontiguousarray(r.cols).reshape(-1), r.OM) == int(hs[i]), i


def test_tagged_strips_random_wide_bands(mz):
    # bands wide AND high -- radii 64-160, long indels (70-300 unshared columns), drifting wide bands, full matrices -- go to the
    # 64-row strips: with the tagged wavefront's arithmetic where the pair is well-formed (MZ_MODE_TSTRIP = 4), exact otherwise
    # (MZ_MODE_STRIP = 1).  Strip boundaries at 63 / 64 / 65 / 127 / 128 / 129 rows; every pair against the oracle by hash, with
    # the default kernels, the wavefront family alone, and the exact kernels.
    from multiz_amd import synth
    rng = np.random.default_rng(44)
    pairs = []
    while len(pairs) < 1500:
        A, B, LB, RB = inputs.random_wide_pair(rng)
        if mo.check(A.shape[0], B.shape[0], LB, RB)[0] == 0:
            pairs.append((A, B, LB, RB))
    batch = synth.pack_pairs(pairs)
    om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=min(64, os.cpu_count() or 8))
    assert bad == 0
    for which in (2, 1, 0):
        _kernels(mz, which)
        db = mz.DevBatch(batch)
        db.run()
        res = db.results()
        assert (res["status"] == 0).all() and cells == int(res["cells"].sum())
        hist = np.bincount(res["mode"], minlength=14)
        if which:
            # the two-wave wavefront (MZ_MODE_DUO = 13) where 128 rows in flight hold the band, the tagged strips beyond (the rolling form is opt-in)
            assert hist[4] + hist[13] > 300 and hist[13] > 50 and hist[4] > 50 and hist[12] == 0, hist
        else:
            assert hist[4] == 0 and hist[12] == 0 and hist[13] == 0 and hist[1] > 300, hist     # exact kernels only
        out = db.out.cpu().numpy()
        for i, (A, B, _, _) in enumerate(pairs):
            m_, o0, W = int(res["om"][i]), int(res["offOut"][i]), A.shape[1] + B.shape[1]
            assert m_ == om[i] and _hash(out[o0: o0 + m_ * W], m_) == int(hs[i]), (which, i, int(res["mode"][i]))
    _kernels(mz, 2)


def test_host_paths_with_every_late_piece_run_twice():
    # MZ_HEDGE_US=1: every piece of the host paths' packing and assembling loops that is not back within a microsecond is handed out a
    # second time (mz_pool.c) -- pieces must give the same result whoever else is at work on their chunk by then.  Round 5 shipped two
    # that did not for a day: a second run of a packing piece read the slot of a pair's band steps from an array the send stage had
    # meanwhile turned into exception-block offsets (a fault, or another pair's steps overwritten), and reset a pair's format byte
    # before setting it again.  The differential sweeps of both paths (bands with nibble, byte and raw steps; chunks of 37 pairs).
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MZ_HEDGE_US="1", MZ_HEDGE_DELAY_US="300", MZ_CHUNK_PAIRS="37")
    for tool, args, what in (("host_sweep.py", ["700", "702"], "pairs"), ("preyama_sweep.py", ["700", "701"], "merges")):
        p = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", tool)] + args, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
        last = p.stdout.strip().splitlines()[-1]
        assert last.startswith(what) and last.endswith("bad 0"), last


def test_rolling_form_with_late_starts_opt_in():
    # MZ_TROLL=1 (read when the score model is uploaded, hence the child): well-formed wide-and-high bands whose rows never wait more
    # than the rings hold run on the tagged wavefront with late starts (MZ_MODE_TROLL = 12, kernels/roll.inc) instead of the strips;
    # 2 000 random wide pairs against the oracle by hash (tests/tools/strip_stress.py)
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "strip_stress.py"), "2000", "7"], cwd=root,
                       env=dict(os.environ, MZ_TROLL="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    last = p.stdout.strip().splitlines()[-1]
    hist = [int(x) for x in last[last.index("[") + 1: last.index("]")].split()]
    assert hist[12] > 500 and "mismatches 0" in last, last


def test_pipelined_form_from_a_cold_start():
    # mz_dev_run_async() as the FIRST call of a process (its helper streams are created on first use) and again after
    # mz_finalize(): results equal the serial form's
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, ".")
import multiz_amd as mz
from multiz_amd import synth
batch = synth.make_batch(400, 2, 2, 200, 400, 30, first_pair=5)
for rep in range(2):
    mz.api.init(0)
    a = mz.DevBatch(batch); b = a.alternate()
    for k in range(4): (a, b)[k % 2].run_async()
    a.wait()
    ra, rb = a.results(), b.results()
    ref = mz.DevBatch(batch); ref.run(); r0 = ref.results()
    for r in (ra, rb):
        assert np.array_equal(r["om"], r0["om"]) and np.array_equal(r["final3"], r0["final3"]) and (r["status"] == 0).all()
    del a, b, ref
    mz.lib().mz_finalize()
print("cold ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, timeout=600)
    assert p.returncode == 0 and b"cold ok" in p.stdout, p.stderr.decode()[-2000:]


def test_score_tables_edited_in_place_are_noticed(mz):
    # the caller's ss / gop are read at call time (mz_scores.h:8-11): a table changed IN PLACE -- same pointers -- must
    # reach the GPU too (a checksum of the class scores, gop and gap_extend decides, not pointer identity)
    lib = mz.lib()
    rng = np.random.default_rng(8)
    pairs = [inputs.make_pair(rng, 2, 2, 150, 160, 30, "diag", mo.smooth) for _ in range(6)]
    lib.init_scores70()
    before = mz.yama_batch(pairs)
    ss = C.POINTER(C.POINTER(C.c_int)).in_dll(lib, "ss")
    gap_extend = C.c_int.in_dll(lib, "gap_extend")
    old = {}
    try:
        for x in b"Aa":
            for y in b"Aa":
                old[(x, y)] = ss[x][y]
                ss[x][y] = 40                                          # A:A 91 -> 40, class structure kept
        gap_extend.value = 45                                          # (the table's dash entries stay -30: the globals are read as they are)
        sc = mo.scores70()                                             # the same edit on the oracle's copy of the tables
        for x in b"Aa":
            for y in b"Aa":
                sc.ss[x][y] = 40
        sc.gap_extend = 45
        after = mz.yama_batch(pairs)
        for p_, r in zip(pairs, after):
            w = mo.yama(*p_, sc=sc)
            assert r.status == 0 and r.OM == w.OM and np.array_equal(r.cols, w.cols) and np.array_equal(r.score, w.final)
        assert sum(not np.array_equal(a.score, b.score) for a, b in zip(before, after)) == len(pairs)
    finally:
        for (x, y), v in old.items():
            ss[x][y] = v
        gap_extend.value = 30
        lib.init_scores70()
    again = mz.yama_batch(pairs)
    for a, b in zip(before, again):
        assert np.array_equal(a.score, b.score) and np.array_equal(a.cols, b.cols)


def test_empty_and_tiny_batches(mz):
    assert mz.yama_batch([]) == []
    A = np.frombuffer(b"A", dtype=np.uint8).reshape(1, 1)
    r = mz.yama_one(A, A, np.array([0, 0], dtype=np.int32), np.array([1, 1], dtype=np.int32))
    assert r.status == 0 and r.OM == 1 and bytes(r.cols.ravel()) == b"AA"


def _check(res, pairs, tags=None, exact_scores=True):
    for i, ((A, B, LB, RB), r) in enumerate(zip(pairs, res)):
        tag = tags[i] if tags else i
        want = mo.yama(A, B, LB, RB, variant="profile")
        assert want.rc == 0
        assert r.status == 0, (tag, r.status)
        assert r.OM == want.OM, tag
        assert np.array_equal(r.cols, want.cols), tag
        # (C,D,I) at (M,N): the fast kernel may hold a different value in an UNREACHABLE state
        # (about -2^30 either way); reachable ones and the winner must agree
        if exact_scores:
            assert np.array_equal(r.score, want.final), tag
        else:
            assert r.score.max() == want.final.max(), tag
            live = want.final > -(1 << 29)
            assert np.array_equal(r.score[live], want.final[live]), tag


def test_band_steps_that_do_not_fit_a_byte(mz):
    # mz_yama_batch() ships LB / RB as byte steps where every step is 0..255 and raw otherwise: a band that jumps by
    # hundreds of columns per row (legal: monotone, connected, wide enough) must take the raw route, next to ordinary
    # pairs in the same call, and give the reference's result
    rng = np.random.default_rng(77)
    pairs = []
    for M, N, step in ((6, 2000, 300), (9, 3000, 320), (40, 1200, 0), (30, 1500, 40), (64, 2600, 37)):   # raw, raw, nibble, byte, byte steps
        A = inputs.random_block(rng, M, 2, dash=0.05, odd=0.02)
        B = inputs.noisy_copy(rng, A, N, 3, dash=0.05)
        if step:
            LB = np.array([0, 0] + [min(step * i, N - 500) for i in range(1, M)], dtype=np.int32)
            RB = np.array([min(step * (i + 1) + 100, N) for i in range(M)] + [N], dtype=np.int32)
            RB = np.maximum.accumulate(RB)
        else:
            LB, RB = mo.smooth(*inputs.diag_band(M, N), M, N, 30)
        assert mo.check(M, N, LB, RB)[0] == 0
        pairs.append((A, B, LB, RB))
    assert max(int(np.diff(p[2]).max()) for p in pairs[:2]) > 255
    assert all(15 < max(int(np.diff(p[2]).max()), int(np.diff(p[3]).max())) < 256 for p in pairs[3:])
    for kset in (2, 1, 0):
        _kernels(mz, kset)
        res = mz.yama_batch(pairs)
        for i, (p, r) in enumerate(zip(pairs, res)):
            w = mo.yama(*p)
            assert r.status == 0 and r.OM == w.OM and np.array_equal(r.cols, w.cols), (kset, i)
    _kernels(mz, 2)


def test_hints_of_another_kernel_selection_are_ignored(mz):
    # DevBatch derives dp_hint / dp_grid / dp_rows / walk_hint from the plan made when it is built; every run re-plans on
    # the device under the selection in force THEN.  Built under "wavefront kernels only" and run after the row-parallel
    # kernels are switched on, the stale hints would launch the wavefront kernels alone while the re-plan hands most
    # pairs to k_dp_row: traceback never written, status MZ_OK, wrong columns.  The hints carry the selection's
    # generation (mz_hint_generation) and are dropped when it differs.  (Pairs whose slices no longer fit the
    # workspaces sized under the old selection -- the transposed pairs need a prep slice -- fail loudly, as before.)
    from multiz_amd import synth
    rng = np.random.default_rng(4242)
    pairs = []
    for _ in range(60):                     # bands no wider than 61 columns, N <= M: row-parallel pairs that need no prep slice
        M = int(rng.integers(80, 400))
        pairs.append(inputs.make_pair(rng, int(rng.integers(1, 4)), int(rng.integers(1, 4)), M, M - int(rng.integers(0, 20)), 30, "diag", mo.smooth))
    batch = synth.pack_pairs(pairs)
    _kernels(mz, 1)
    db = mz.DevBatch(batch)                 # workspaces and hints of the wavefront kernels (the larger traceback)
    stamp = db.c.hint_gen
    _kernels(mz, 2)
    db.run()
    assert mz.lib().mz_hint_generation() != stamp
    for form in ("serial", "pipelined"):
        res = db.results()
        ok = res["status"] == 0
        assert set(np.unique(res["status"])) <= {0, 19}, np.unique(res["status"])
        assert int((ok & (res["mode"] == 5)).sum()) >= 40, (np.bincount(res["mode"]), np.unique(res["status"]))   # row-parallel pairs the stale hint would skip
        host_out = db.out.cpu().numpy()
        for i, (A, B, LB, RB) in enumerate(pairs):
            if not ok[i]:
                continue
            want = mo.yama(A, B, LB, RB)
            m_, o0 = int(res["om"][i]), int(res["offOut"][i])
            assert m_ == want.OM and np.array_equal(host_out[o0: o0 + m_ * (A.shape[1] + B.shape[1])].reshape(m_, -1), want.cols), (form, i, int(res["mode"][i]))
        db.out.zero_(); db.tbw.zero_()
        db.run_async(); db.wait()


def test_host_batches_are_chunked(tmp_path):
    # mz_yama_batch() runs big batches in chunks through the same staging buffers (MZ_CHUNK_PAIRS=7 here, read
    # once per process, hence the subprocess): results must not depend on the chunking
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import inputs
import multiz_amd as mz
from oracle import mzoracle as mo
mz.api.init(0)
rng = np.random.default_rng(99)
pairs = []
while len(pairs) < 40:
    K, L, M, N = int(rng.integers(1, 6)), int(rng.integers(1, 6)), int(rng.integers(20, 300)), int(rng.integers(20, 300))
    p = inputs.make_pair(rng, K, L, M, N, 30, "diag", mo.smooth)
    if mo.check(M, N, p[2], p[3])[0] == 0:
        pairs.append(p)
pairs.insert(13, (pairs[0][0], pairs[0][1], pairs[0][2][::-1].copy(), pairs[0][3]))     # one illegal band in the middle
res = mz.yama_batch(pairs)
for i, (p, r) in enumerate(zip(pairs, res)):
    if i == 13:
        assert r.status != 0
        continue
    w = mo.yama(*p)
    assert r.status == 0 and r.OM == w.OM and np.array_equal(r.cols, w.cols), i
print("chunked ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MZ_CHUNK_PAIRS="7")
    p = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, timeout=600)
    assert p.returncode == 0 and b"chunked ok" in p.stdout, p.stderr.decode()[-2000:]


@pytest.mark.parametrize("walk", ["wave", "direct", "reg"])
def test_every_walk_kernel(walk):
    # three traceback walks -- the run-following one with its window in LDS, the step-by-step chase (MZ_WALK=direct; read once per
    # process, hence the subprocess) and the run-following one with its window in registers (MZ_WALK=reg: the row-parallel pairs; the
    # others by the LDS form) which the walks beside DP kernels take: each must produce the reference's merged blocks for every DP layout -- row-parallel ROW / COL
    # (lift and rotate forms), tagged and untagged wavefronts, the strip kernel -- on short and on long pairs
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import inputs
import multiz_amd as mz
from oracle import mzoracle as mo
mz.api.init(0)
rng = np.random.default_rng(2024)
shapes = [(2, 2, 300, 310, 30, "diag"), (3, 2, 40, 700, 30, "diag"), (2, 3, 700, 40, 30, "diag"), (2, 2, 2500, 2400, 30, "diag"),
          (4, 4, 150, 150, 12, "wander"), (2, 2, 230, 150, 25, "wander"), (2, 2, 90, 90, 60, "diag"), (1, 1, 30, 33, 30, "full"),
          (29, 29, 200, 210, 30, "diag"), (2, 2, 5000, 5100, 30, "diag")]
pairs = []
for K, L, M, N, R, band in shapes * 2:
    p = inputs.make_pair(rng, K, L, M, N, R, band, mo.smooth)
    if mo.check(M, N, p[2], p[3])[0] == 0:
        pairs.append(p)
for K, L, M, N, R, band in ((2, 2, 400, 380, 30, "diag"), (3, 1, 200, 600, 40, "wander"), (1, 1, 120, 100, 30, "full")):
    p = inputs.make_pair(rng, K, L, M, N, R, band, mo.smooth)       # unrelated sequences: the path turns all the time
    p = (p[0], inputs.random_block(rng, N, L, dash=0.3, odd=0.05), p[2], p[3])
    if mo.check(M, N, p[2], p[3])[0] == 0:
        pairs.append(p)
modes = set()
for fast, row in ((1, 1), (1, 0), (0, 0)):
    mz.lib().mz_enable_fast(fast); mz.lib().mz_enable_row(row)
    res = mz.yama_batch(pairs)
    for i, (p, r) in enumerate(zip(pairs, res)):
        w = mo.yama(*p)
        assert r.status == 0 and r.OM == w.OM and np.array_equal(r.cols, w.cols), (fast, row, i)
print("walk ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MZ_WALK=walk)
    p = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, timeout=900)
    assert p.returncode == 0 and b"walk ok" in p.stdout, p.stderr.decode()[-2000:]


def test_host_path_sweep(mz):
    # mz_yama_batch() end to end on shapes that take every branch of its host half: thin blocks (four columns per
    # step by byte shuffle), blocks of up to 4+4 rows (fixed-size moves), wider ones (16-byte copies); band bounds whose
    # steps need the nibble, the byte and the raw format in one call; pairs the plan refuses in between (the
    # reference's own status, nothing assembled for them).  tests/tools/host_sweep.py is the long form.
    rng = np.random.default_rng(910_000)
    pairs, want_bad = [], []
    while len(pairs) < 500:
        kind = int(rng.integers(0, 7))
        K, L = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        M, N = int(rng.integers(1, 600)), int(rng.integers(1, 600))
        R = int(rng.choice([3, 10, 30, 31, 64]))
        band = str(rng.choice(["diag", "wander", "wander", "full"]))
        if kind == 0:
            K, L = int(rng.integers(5, 40)), int(rng.integers(5, 40)); M, N = int(rng.integers(5, 200)), int(rng.integers(5, 200))
        elif kind == 1:
            K, L = int(rng.integers(1, 3)), int(rng.integers(1, 3)); M, N = int(rng.integers(800, 3000)), int(rng.integers(800, 3000)); band = "diag"
        A, B, LB, RB = inputs.make_pair(rng, K, L, M, N, R, band, mo.smooth,
                                        dash=float(rng.choice([0.0, 0.08, 0.5])), odd=float(rng.choice([0.0, 0.05, 0.6])))
        if band == "full" and M * N > 60000:
            continue
        if kind == 2 and M > 8:                      # a jump of tens / hundreds of columns in the bounds
            j = int(rng.integers(2, M - 2)); jump = int(rng.choice([20, 90, 300, 700]))
            if N > jump + 40:
                LB = LB.copy(); RB = RB.copy()
                RB[j:] = np.minimum(RB[j:] + jump, N); LB[j + 1:] = np.minimum(LB[j + 1:] + jump, N - 11 if N > 11 else 0)
                LB = np.maximum.accumulate(LB); RB = np.maximum.accumulate(RB); RB[-1] = N
        rc = mo.check(M, N, LB, RB)[0]
        if rc != 0 and rng.random() < 0.9:
            continue
        pairs.append((A, B, LB.astype(np.int32), RB.astype(np.int32))); want_bad.append(rc)
    assert sum(1 for r in want_bad if r) >= 3
    _kernels(mz, 2)
    res = mz.yama_batch(pairs)
    for i, ((A, B, LB, RB), r) in enumerate(zip(pairs, res)):
        if want_bad[i]:
            assert r.status == want_bad[i] and r.cols is None, i
            continue
        w = mo.yama(A, B, LB, RB, variant="profile")
        assert r.status == 0 and r.OM == w.OM and np.array_equal(r.cols, w.cols), (i, A.shape, B.shape)


def test_below_the_sentinel_is_reported_as_such(mz):
    """Two blocks of 100+ rows EACH over hundreds of mismatching columns: scores fall below the reference's MININT sentinel (mz_yama.c:29),
    "unreachable" states win its comparisons, and its walk steps outside the band, reads neighbouring rows' traceback bytes and still
    arrives (tests/golden/below_sentinel.npz: the compiled reference's OM and column hash, which the oracle reproduces).  The product
    does not follow it there: such a pair comes back MZ_E_SENTINEL -- not MZ_E_TRACEBACK, the reference's own error, and never another
    alignment -- on every kernel set, through the device-resident API and through yama_batch()."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "below_sentinel.npz"))
    A, B, LB, RB = z["A"], z["B"], z["LB"], z["RB"]
    K, L, M, N = A.shape[1], B.shape[1], A.shape[0], B.shape[0]
    assert K * L * 525 * (M + N + 2) >= 1 << 30                  # scores CAN get below -2^30 here
    w = mo.yama(A, B, LB, RB, variant="profile")                 # (the oracle follows the reference: a result, the fixture's)
    assert w.rc == 0 and w.OM == int(z["OM"]) and mo.fnv1a_np(w.cols, mo.fnv1a_np(np.array([w.OM], dtype=np.int32).view(np.uint8))) == int(z["hash"])
    ok = inputs.make_pair(np.random.default_rng(5), 2, 2, 120, 130, 30, "diag", mo.smooth)
    for fast, row in ((1, 1), (1, 0), (0, 0)):
        mz.lib().mz_enable_fast(fast); mz.lib().mz_enable_row(row)
        res = mz.yama_batch([ok, (A, B, LB, RB), ok])
        assert [r.status for r in res] == [0, 21, 0] and res[1].cols is None, [r.status for r in res]
        assert mz.api.MZ_STATUS[res[1].status] == "sentinel"
    mz.lib().mz_enable_fast(1); mz.lib().mz_enable_row(1)


def test_rebased_kernels_report_a_frontier_below_the_sentinel(mz):
    """The row-parallel kernels re-base their scores and would find the TRUE optimum however low the scores go; the reference does not (its
    sentinel states start winning below -2^30).  Two blocks of 30 rows over thousands of UNRELATED columns: at 3 000 columns the scores stay
    above the sentinel and the result is the reference's; at 14 000 the frontier falls below -2^29 on the way, the kernels mark the pair
    (kernels/row.inc: ROW_SENTINEL_EDGE) and it comes back MZ_E_SENTINEL -- or, where the plan took an exact kernel, as the oracle has it."""
    rng = np.random.default_rng(77)
    pairs = []
    for n in (3000, 14000):
        A = inputs.random_block(rng, n, 30, dash=0.05, odd=0.0)
        B = inputs.random_block(rng, n, 30, dash=0.05, odd=0.0)
        LB, RB = inputs.diag_band(n, n)
        LB, RB = mo.smooth(LB, RB, n, n, 30)
        pairs.append((A, B, LB.astype(np.int32), RB.astype(np.int32)))
    for row in (1, 0):
        mz.lib().mz_enable_fast(1); mz.lib().mz_enable_row(row)
        res = mz.yama_batch(pairs)
        for i, (p, r) in enumerate(zip(pairs, res)):
            w = mo.yama(*p, variant="profile")
            same = r.status == 0 and r.OM == w.OM and np.array_equal(r.cols, w.cols)
            assert same or (i == 1 and r.status == 21), (row, i, r.status)
        assert res[0].status == 0
    mz.lib().mz_enable_fast(1); mz.lib().mz_enable_row(1)
