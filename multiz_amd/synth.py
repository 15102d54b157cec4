"""Synthetic block-pair batches in the packed layout of include/mz_amd.h.
Thin binding of mz_synth_shapes()/mz_synth_fill() in libmzamd.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .api import lib

BASE_SEED = 88172645463325252

# BASELINE.json configs -> (K, L, columns lo, columns hi, pairs, radius)
CONFIGS = {
    "c1": dict(K=2, L=1, mlo=180, mhi=220, pairs=100, radius=30),
    "c2": dict(K=2, L=2, mlo=900, mhi=1100, pairs=50000, radius=30),
    "c3": dict(K=10, L=10, mlo=1800, mhi=2200, pairs=5000, radius=30),
    # configs[3]: 1 M pairs over 8 GPUs from a 30-leaf caterpillar+balanced guide tree (mz_synth.c: tree30); K = L = 0
    # stands for "K, L of a random tree node per pair"; `pairs` is one GPU's share
    "c4": dict(K=0, L=0, mlo=200, mhi=1000, pairs=125000, radius=30),
    "c5": dict(K=2, L=2, mlo=95000, mhi=105000, pairs=1000, radius=30),
    # not a BASELINE configuration: C2's blocks with the bands pre_yama derives when the blocks have indels against the
    # shared reference row, 10 runs of unshared columns per 1 000 (mean length 3) -- rows wider than 64 columns
    "c2i": dict(K=2, L=2, mlo=900, mhi=1100, pairs=20000, radius=30, indel=10),
    # not BASELINE configurations: C2's blocks with bands too wide AND too high for the row-parallel kernels -- radius 50 (rows of
    # 101 columns: the tagged anti-diagonal wavefront, MZ_MODE_FASTT) and radius 100 (rows of 201: 64-row strips, MZ_MODE_STRIP);
    # the fallback kernels' rates for the record
    "c2w": dict(K=2, L=2, mlo=900, mhi=1100, pairs=20000, radius=50),
    "c2s": dict(K=2, L=2, mlo=900, mhi=1100, pairs=10000, radius=100),
    # not a BASELINE configuration: c2i's bands with a heavy tail -- 20 of 1 000 runs are 70-300 columns long (one such run makes the
    # rows around it too wide for the row-parallel kernels and the anti-diagonals too high for the rolling wavefront)
    "c2g": dict(K=2, L=2, mlo=900, mhi=1100, pairs=20000, radius=30, indel=10 | (20 << 16)),
    # ... and the guide-tree workload's blocks (1..29 rows) with such bands
    "c4i": dict(K=0, L=0, mlo=200, mhi=1000, pairs=50000, radius=30, indel=10),
}


def describe(name: str, pairs: int) -> str:
    c = CONFIGS[name]
    rows = "K,L from a 30-leaf caterpillar+balanced guide tree (1..29 rows)" if c["K"] == 0 else f"{c['K']}+{c['L']} rows"
    if c.get("indel"):
        ev, tail = c["indel"] & 0xffff, c["indel"] >> 16
        return (f"{name}: {pairs} block pairs/GPU, {rows}, M~U[{c['mlo']},{c['mhi']}], bands of blocks with {ev} indel runs "
                f"per 1000 columns (mean length 3" + (f"; {tail} of 1000 runs 70-300 columns long" if tail else "") + f"), R={c['radius']}")
    return f"{name}: {pairs} block pairs/GPU, {rows}, M,N~U[{c['mlo']},{c['mhi']}], diag band R={c['radius']}"


def tree_nodes():
    """(K, L) of the 29 internal nodes of the C4 guide tree"""
    l = lib()
    K, L = np.zeros(29, dtype=np.int32), np.zeros(29, dtype=np.int32)
    l.mz_synth_tree_nodes.argtypes = [C.c_void_p, C.c_void_p]
    assert l.mz_synth_tree_nodes(K.ctypes.data, L.ctypes.data) == 29
    return K, L


def make_batch(n: int, K: int, L: int, mlo: int, mhi: int, radius: int = 30, seed: int = BASE_SEED,
               first_pair: int = 0, indel: int = 0) -> dict:
    l = lib()
    aK, aL, aM, aN = (np.zeros(n, dtype=np.int32) for _ in range(4))
    oA, oB, oBand = (np.zeros(n, dtype=np.int64) for _ in range(3))
    tot = (C.c_int64 * 3)()
    if K == 0 and L == 0 and not indel:  # the tree workload (configs[3])
        l.mz_synth_shapes_tree.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_void_p]
        l.mz_synth_shapes_tree(n, seed, first_pair, mlo, mhi, aK.ctypes.data, aL.ctypes.data, aM.ctypes.data, aN.ctypes.data,
                               oA.ctypes.data, oB.ctypes.data, oBand.ctypes.data, C.cast(tot, C.c_void_p))
    elif indel:
        l.mz_synth_shapes_indel.argtypes = [C.c_int, C.c_uint64, C.c_int64] + [C.c_int] * 5 + [C.c_void_p] * 7 + [C.c_void_p]
        l.mz_synth_shapes_indel(n, seed, first_pair, K, L, mlo, mhi, indel, aK.ctypes.data, aL.ctypes.data, aM.ctypes.data,
                                aN.ctypes.data, oA.ctypes.data, oB.ctypes.data, oBand.ctypes.data, C.cast(tot, C.c_void_p))
    else:
        l.mz_synth_shapes.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_void_p]
        l.mz_synth_shapes(n, seed, first_pair, K, L, mlo, mhi, aK.ctypes.data, aL.ctypes.data, aM.ctypes.data, aN.ctypes.data,
                          oA.ctypes.data, oB.ctypes.data, oBand.ctypes.data, C.cast(tot, C.c_void_p))
    poolA = np.zeros(max(tot[0], 1), dtype=np.uint8)
    poolB = np.zeros(max(tot[1], 1), dtype=np.uint8)
    poolLB = np.zeros(max(tot[2], 1), dtype=np.int32)
    poolRB = np.zeros(max(tot[2], 1), dtype=np.int32)
    l.mz_synth_fill.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int] + [C.c_void_p] * 11
    l.mz_synth_fill(n, seed, first_pair, radius, aK.ctypes.data, aL.ctypes.data, aM.ctypes.data, aN.ctypes.data,
                    oA.ctypes.data, oB.ctypes.data, oBand.ctypes.data,
                    poolA.ctypes.data, poolB.ctypes.data, poolLB.ctypes.data, poolRB.ctypes.data)
    if indel:
        l.mz_synth_bands_indel.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 5
        l.mz_synth_bands_indel(n, seed, first_pair, radius, indel, aM.ctypes.data, aN.ctypes.data, oBand.ctypes.data,
                               poolLB.ctypes.data, poolRB.ctypes.data)
    return dict(K=aK, L=aL, M=aM, N=aN, offA=oA, offB=oB, offBand=oBand,
                poolA=poolA, poolB=poolB, poolLB=poolLB, poolRB=poolRB)


def pair_of(batch: dict, i: int):
    K, L, M, N = (int(batch[k][i]) for k in ("K", "L", "M", "N"))
    a0, b0, d0 = int(batch["offA"][i]), int(batch["offB"][i]), int(batch["offBand"][i])
    A = batch["poolA"][a0: a0 + K * M].reshape(M, K)
    B = batch["poolB"][b0: b0 + L * N].reshape(N, L)
    return A, B, batch["poolLB"][d0: d0 + M + 1], batch["poolRB"][d0: d0 + M + 1]


def subset(batch: dict, idx) -> dict:
    """re-pack the chosen pairs into a new batch (used for the CPU-baseline sample)"""
    from .shard import take
    return take(batch, np.asarray(idx, dtype=np.int64))


def band_cells(batch: dict) -> int:
    return int((batch["poolRB"].astype(np.int64) - batch["poolLB"].astype(np.int64) + 1)[: int(batch["offBand"][-1]) + int(batch["M"][-1]) + 1].sum())


def pack_pairs(pairs) -> dict:
    """a batch in the packed-pool layout from a list of (A, B, LB, RB) numpy tuples (A: (M,K) uint8, B: (N,L))"""
    n = len(pairs)
    out = {k: np.zeros(n, dtype=np.int32) for k in ("K", "L", "M", "N")}
    oa, ob, od, pa, pb, plb, prb = [], [], [], [], [], [], []
    a = b = d = 0
    for i, (A, B, LB, RB) in enumerate(pairs):
        out["M"][i], out["K"][i] = A.shape
        out["N"][i], out["L"][i] = B.shape
        oa.append(a); ob.append(b); od.append(d)
        pa.append(np.ascontiguousarray(A, dtype=np.uint8).ravel()); pb.append(np.ascontiguousarray(B, dtype=np.uint8).ravel())
        plb.append(np.asarray(LB, dtype=np.int32)); prb.append(np.asarray(RB, dtype=np.int32))
        a += A.size; b += B.size; d += len(LB)
    out.update(offA=np.array(oa, dtype=np.int64), offB=np.array(ob, dtype=np.int64), offBand=np.array(od, dtype=np.int64),
               poolA=np.concatenate(pa), poolB=np.concatenate(pb), poolLB=np.concatenate(plb), poolRB=np.concatenate(prb))
    return out


# ------------------------------------------------------------------------------------------ block TEXT (mz_preyama_batch)

PREJOB_DT = np.dtype({"names": ["K", "L1", "M_all", "N_all", "radius", "rows1", "rows2", "v"],
                      "formats": ["<i4", "<i4", "<i4", "<i4", "<i4", "<u8", "<u8", "<i4"],
                      "offsets": [0, 4, 8, 12, 16, 24, 32, 40], "itemsize": 48})
PREOUT_DT = np.dtype({"names": ["status", "badrow", "null_result", "stage", "M", "N", "OM", "score", "size", "rows", "block"],
                      "formats": ["<i4"] * 7 + ["<f8", "<u8", "<u8", "<u8"],
                      "offsets": [0, 4, 8, 12, 16, 20, 24, 32, 40, 48, 56], "itemsize": 64})


def make_pre_batch(n: int, K: int, L: int, rlo: int, rhi: int, radius: int = 30, events: int = 2, v: int = 1,
                   seed: int = BASE_SEED, first_pair: int = 0) -> dict:
    """n merges as mz_prejob records over one text pool (mz_synth_pre_shapes / mz_synth_pre_fill): block 1 of K rows, block 2 of
    L rows below its copy of the shared reference row, R ~ U[rlo, rhi] reference bases, `events` runs of unshared columns per
    1 000 bases.  v: 1 one-stage merges, 0 two-stage (merges whose first block has a single row stay one-stage), 2 alternating.
    The returned dict keeps everything the records point into alive."""
    from .api import PreJob, PreOut
    l = lib()
    assert PREJOB_DT.itemsize == C.sizeof(PreJob) and PREOUT_DT.itemsize == C.sizeof(PreOut)
    aK, aL1, aMa, aNa = (np.zeros(n, dtype=np.int32) for _ in range(4))
    off = np.zeros(n, dtype=np.int64)
    tot = C.c_int64(0)
    l.mz_synth_pre_shapes.argtypes = [C.c_int, C.c_uint64, C.c_int64] + [C.c_int] * 5 + [C.c_void_p] * 5 + [C.c_void_p]
    l.mz_synth_pre_shapes(n, seed, first_pair, K, L, rlo, rhi, events, aK.ctypes.data, aL1.ctypes.data, aMa.ctypes.data, aNa.ctypes.data,
                          off.ctypes.data, C.cast(C.byref(tot), C.c_void_p))
    pool = np.zeros(max(tot.value, 1) + 64, dtype=np.uint8)
    l.mz_synth_pre_fill.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
    l.mz_synth_pre_fill(n, seed, first_pair, rlo, rhi, events, aK.ctypes.data, aL1.ctypes.data, aMa.ctypes.data, aNa.ctypes.data,
                        off.ctypes.data, pool.ctypes.data)
    # row pointers: K of block 1, then L1 of block 2, merge after merge
    nptr = aK.astype(np.int64) + aL1
    pstart = np.concatenate(([0], np.cumsum(nptr)))
    ptrs = np.zeros(int(pstart[-1]), dtype=np.uint64)
    base = pool.ctypes.data + off
    which = np.repeat(np.arange(n), nptr)
    k = np.arange(int(pstart[-1])) - pstart[which]
    in1 = k < aK[which]
    ptrs[:] = np.where(in1, base[which] + k * aMa[which].astype(np.int64),
                       base[which] + aK[which].astype(np.int64) * aMa[which] + (k - aK[which]) * aNa[which].astype(np.int64)).astype(np.uint64)
    jobs = np.zeros(n, dtype=PREJOB_DT)
    jobs["K"], jobs["L1"], jobs["M_all"], jobs["N_all"], jobs["radius"] = aK, aL1, aMa, aNa, radius
    jobs["rows1"] = ptrs.ctypes.data + 8 * pstart[:-1].astype(np.uint64)
    jobs["rows2"] = ptrs.ctypes.data + 8 * (pstart[:-1] + aK).astype(np.uint64)
    vv = np.full(n, 1, dtype=np.int32) if v == 1 else np.zeros(n, dtype=np.int32) if v == 0 else (np.arange(n) % 2).astype(np.int32)
    vv[aK < 2] = 1
    jobs["v"] = vv
    return dict(jobs=jobs, outs=np.zeros(n, dtype=PREOUT_DT), pool=pool, ptrs=ptrs, off=off, K=aK, L1=aL1, Ma=aMa, Na=aNa)


def pre_rows_of(pb: dict, i: int):
    """(rows1, rows2) of merge i as lists of bytes"""
    K, L1, Ma, Na, o = int(pb["K"][i]), int(pb["L1"][i]), int(pb["Ma"][i]), int(pb["Na"][i]), int(pb["off"][i])
    t = pb["pool"]
    r1 = [t[o + k * Ma: o + (k + 1) * Ma].tobytes() for k in range(K)]
    o2 = o + K * Ma
    r2 = [t[o2 + k * Na: o2 + (k + 1) * Na].tobytes() for k in range(L1)]
    return r1, r2
