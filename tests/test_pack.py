"""CPU-side checks of the host half of mz_yama_batch()'s link formats (multiz_amd/csrc/mz_pack.c): byte classes two
per byte, band bounds as nibble / byte steps, and the merged columns assembled from a 2-bit edit script -- against
numpy restatements and against the oracle's merged columns (reference mz_yama.c:293-313).  The device halves
(k_unnib, k_unband, k_script_pack) are covered by the -m gpu parity tests, which go through the same path."""
import ctypes as C
import os

import numpy as np
import pytest

import inputs
from oracle import mzoracle as mo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(os.path.join(ROOT, "multiz_amd", "libmzamd.so"))
    L.mz_pack_classes.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.mz_pack_band_nib.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.mz_pack_band_nib.restype = C.c_uint32
    L.mz_pack_band_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.mz_assemble_cols.argtypes = [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_int, C.c_void_p]
    L.mz_pack_classes_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.mz_pack_band_nib_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.mz_pack_band_nib_stream.restype = C.c_uint32
    return L


def aligned(nbytes: int, align: int = 64, phase: int = 0) -> np.ndarray:
    """uint8 view of nbytes whose address is `phase` past a multiple of `align`"""
    raw = np.full(nbytes + 2 * align, 0xEE, dtype=np.uint8)
    off = (-raw.ctypes.data) % align + phase
    return raw[off: off + nbytes]


def class_of(x: np.ndarray) -> np.ndarray:
    """the six byte classes of the score tables (reference mz_scores.c:39-54), as the device's byte_class()"""
    t = np.full(256, 5, dtype=np.uint8)
    for i, ch in enumerate("ACGT"):
        t[ord(ch)] = t[ord(ch.lower())] = i
    t[ord("-")] = 4
    return t[x]


@pytest.mark.parametrize("n", [0, 1, 2, 7, 63, 64, 65, 127, 128, 1000, 4097])
def test_class_nibbles(lib, n):
    rng = np.random.default_rng(n)
    src = rng.integers(0, 256, size=n, dtype=np.uint8)
    letters = np.frombuffer(b"ACGTacgt-NnX\r\x00\xff", dtype=np.uint8)
    mask = rng.random(n) < 0.8
    src[mask] = letters[rng.integers(0, len(letters), size=int(mask.sum()))]
    dst = np.full((n + 1) // 2 + 8, 0xEE, dtype=np.uint8)
    lib.mz_pack_classes(src.ctypes.data, n, dst.ctypes.data)
    cls = class_of(src)
    if n % 2:
        cls = np.append(cls, 5)
    want = (cls[0::2] | (cls[1::2] << 4)).astype(np.uint8)
    assert np.array_equal(dst[: (n + 1) // 2], want)
    assert (dst[(n + 1) // 2:] == 0xEE).all()              # nothing written past the end
    # the streaming form: a 32-byte aligned slot, padded with class "other"
    slot = (n + 63) // 64 * 32
    d2 = aligned(slot + 32, 32)
    lib.mz_pack_classes_stream(src.ctypes.data, n, d2.ctypes.data, slot)
    assert np.array_equal(d2[: n // 2], want[: n // 2]) and (d2[slot:] == 0xEE).all()
    if n % 2:
        assert d2[n // 2] == want[n // 2]
    assert (d2[(n + 1) // 2: slot] == 0x55).all()


@pytest.mark.parametrize("M", [1, 2, 15, 16, 17, 100, 1001, 1024, 1025, 2500])
def test_band_steps(lib, M):
    rng = np.random.default_rng(M)
    for hi in (3, 15, 16, 200, 256, 5000):
        LB = np.concatenate([[0], np.cumsum(rng.integers(0, hi + 1, size=M))]).astype(np.int32)
        RB = (LB + rng.integers(0, 40, size=M + 1)).astype(np.int32)
        RB = np.maximum.accumulate(RB).astype(np.int32)
        dl, dr = np.diff(LB), np.diff(RB)
        dst = np.full(8 + M + 8, 0xEE, dtype=np.uint8)
        acc = lib.mz_pack_band_nib(LB.ctypes.data, RB.ctypes.data, M, dst.ctypes.data)
        assert (acc < 16) == bool((dl < 16).all() and (dr < 16).all() and (dl >= 0).all() and (dr >= 0).all())
        assert (acc < 256) == bool((dl < 256).all() and (dr < 256).all())
        assert dst[:8].view(np.int32).tolist() == [int(LB[0]), int(RB[0])] and (dst[8 + M:] == 0xEE).all()
        if acc < 16:
            assert np.array_equal(dst[8: 8 + M], (dl | (dr << 4)).astype(np.uint8))
        slot = (M + 31) // 32 * 32
        d3 = aligned(slot + 32, 32)
        assert lib.mz_pack_band_nib_stream(LB.ctypes.data, RB.ctypes.data, M, d3.ctypes.data, slot) == acc
        assert (d3[slot:] == 0xEE).all() and (d3[M: slot] == 0).all()
        if acc < 16:
            assert np.array_equal(d3[:M], (dl | (dr << 4)).astype(np.uint8))
        if acc < 256:
            d2 = np.full(8 + 2 * M + 8, 0xEE, dtype=np.uint8)
            lib.mz_pack_band_bytes(LB.ctypes.data, RB.ctypes.data, M, d2.ctypes.data)
            assert np.array_equal(d2[8: 8 + M], dl.astype(np.uint8)) and np.array_equal(d2[8 + M: 8 + 2 * M], dr.astype(np.uint8))
            assert (d2[8 + 2 * M:] == 0xEE).all()
    # a step down (an invalid band: the plan must see it as it is) shows as a value >= 256
    LB = np.arange(M + 1, dtype=np.int32); RB = LB + 20
    if M >= 2:
        LB[M // 2 + 1] = LB[M // 2] - 1
        dst = np.zeros(8 + M + 8, dtype=np.uint8)
        assert lib.mz_pack_band_nib(LB.ctypes.data, RB.ctypes.data, M, dst.ctypes.data) >= 256


def script_of(cols: np.ndarray, K: int) -> np.ndarray:
    """edit script of a merged block whose source columns all hold a base (inputs.random_block): I = dashes on top"""
    top = (cols[:, :K] == ord("-")).all(axis=1)
    bot = (cols[:, K:] == ord("-")).all(axis=1)
    assert not (top & bot).any()
    return np.where(top, 1, np.where(bot, 2, 0)).astype(np.uint8)          # FLAG_C 0, FLAG_I 1, FLAG_D 2 (mz_yama.c:24-26)


def pack2(ops: np.ndarray) -> np.ndarray:
    pad = np.zeros((-len(ops)) % 4, dtype=np.uint8)
    o = np.concatenate([ops, pad]).reshape(-1, 4)
    return (o[:, 0] | (o[:, 1] << 2) | (o[:, 2] << 4) | (o[:, 3] << 6)).astype(np.uint8)


@pytest.mark.parametrize("K,L", [(1, 1), (2, 2), (1, 2), (2, 1), (1, 3), (4, 4), (3, 2), (5, 2), (10, 10), (2, 17), (29, 1), (40, 33)])
def test_assemble_matches_the_oracle(lib, K, L):
    rng = np.random.default_rng(K * 100 + L)
    for M, N in ((1, 1), (3, 9), (60, 47), (180, 200)):
        A = inputs.random_block(rng, M, K)
        B = inputs.noisy_copy(rng, A, N, L)
        LB, RB = inputs.diag_band(M, N)
        LB, RB = mo.smooth(LB, RB, M, N, 12)
        want = mo.yama(A, B, LB, RB, variant="profile")
        assert want.rc == 0
        ops = script_of(want.cols, K)
        assert int((ops != 1).sum()) == M and int((ops != 2).sum()) == N
        script = pack2(ops)
        nb = want.OM * (K + L)
        for phase in (0, 16, 37, 63):                       # where the pair's columns start within a 64-byte line
            buf = aligned(nb + 128, 64, 0)
            out = buf[phase: phase + nb + 64]
            lib.mz_assemble_cols(K, L, M, N, A.ctypes.data, B.ctypes.data, script.ctypes.data, want.OM, out.ctypes.data)
            assert np.array_equal(out[:nb].reshape(want.OM, K + L), want.cols), (K, L, M, N, phase)
            assert (out[nb:] == 0xEE).all() and (buf[:phase] == 0xEE).all()     # neighbours' bytes untouched


# ---------------------------------------------------------------------------------------- merged rows (mz_preyama_batch)

class RowSpec(C.Structure):
    _fields_ = [("src", C.c_void_p), ("n", C.c_int), ("squeeze", C.c_int), ("keep", C.c_void_p), ("ops", C.c_void_p), ("tmp", C.c_void_p)]


def bits_to_words(bits: np.ndarray) -> np.ndarray:
    """bool per column -> uint64 words, bit c & 63 of word c >> 6 (the layout k_fin writes)"""
    pad = np.zeros((len(bits) + 63) // 64 * 64, dtype=np.uint8)
    pad[: len(bits)] = bits
    return np.packbits(pad.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint64).copy()


def make_rows(rng, om, nrows, squeeze):
    """random source rows whose masks take exactly their bytes; returns (specs keep-alive, expected rows)"""
    letters = np.frombuffer(b"ACGTacgtNn-", dtype=np.uint8)
    keep, want = [], []
    specs = (RowSpec * nrows)()
    for r in range(nrows):
        ops = rng.random(om) < rng.choice([0.05, 0.5, 0.95, 1.0])
        taken = int(ops.sum())
        mode = squeeze if squeeze != 3 else int(rng.integers(0, 3))
        if mode == 0:
            src = letters[rng.integers(0, len(letters), size=taken)]
            kept, kw = src, None
        elif mode == 1:
            n = taken + int(rng.integers(0, 40))
            kb = np.zeros(n, dtype=bool)
            kb[rng.choice(n, size=taken, replace=False)] = True
            src = letters[rng.integers(0, len(letters), size=n)]
            kept, kw = src[kb], bits_to_words(kb)
        else:
            n = taken + int(rng.integers(0, 40))
            src = np.full(n, ord("-"), dtype=np.uint8)
            pos = np.sort(rng.choice(n, size=taken, replace=False))
            src[pos] = letters[rng.integers(0, 10, size=taken)]          # (no dash among the bases)
            kept, kw = src[pos], None
        src = np.ascontiguousarray(src)
        ow = bits_to_words(ops)
        tmp = np.zeros(len(src) + 16, dtype=np.uint8)
        row = np.full(om, ord("-"), dtype=np.uint8)
        row[ops] = kept
        want.append(row)
        keep += [src, kw, ow, tmp]
        specs[r].src, specs[r].n, specs[r].squeeze = src.ctypes.data, len(src), mode
        specs[r].keep = kw.ctypes.data if kw is not None else None
        specs[r].ops, specs[r].tmp = ow.ctypes.data, tmp.ctypes.data
    return specs, keep, want


@pytest.mark.parametrize("om", [1, 7, 15, 16, 17, 63, 64, 65, 1000, 2048, 2049, 5000])
@pytest.mark.parametrize("squeeze", [0, 1, 2, 3])
def test_assemble_rows(lib, om, squeeze):
    """rows of a merged block from source bytes + masks (mz_assemble_rows): every spread / squeeze form against numpy, at
    every phase of the output within a 64-byte line, nothing written outside the block"""
    lib.mz_assemble_rows.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(om * 4 + squeeze)
    for nrows in (1, 2, 5):
        specs, keep, want = make_rows(rng, om, nrows, squeeze)
        for phase in (0, 1, 31, 63):
            out = aligned(nrows * om + 128, 64, phase)
            out[:] = 0xEE
            lib.mz_assemble_rows(nrows, C.cast(specs, C.c_void_p), om, out[64:].ctypes.data)
            got = out[64: 64 + nrows * om].reshape(nrows, om)
            assert np.array_equal(got, np.stack(want)), (om, squeeze, nrows, phase)
            assert (out[:64] == 0xEE).all() and (out[64 + nrows * om:] == 0xEE).all()
