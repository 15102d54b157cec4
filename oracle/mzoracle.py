"""ctypes front-end for the CPU oracle (oracle/liboracle.so) and, where it has been built,
the compiled reference (oracle/_ref/libref.so).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py -- never by anything under multiz_amd/.

Parity status: PINNED (tests/test_oracle_vs_reference.py against libref.so here, and the
committed tests/golden/ vectors everywhere).

Also holds the Python restatement of the pre_yama() adapter (reference mz_preyama.c:152-359)
used to check the product's C implementation; it calls the C oracle for the DP itself.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libref.so")
REF_O0_PATH = os.path.join(HERE, "_ref", "libref_O0.so")
REF_MULTIZ = os.path.join(HERE, "_ref", "multiz_ref")

NEG = -1073741824
ERRORS = {1: "termination", 2: "narrow", 3: "lb_mono", 4: "rb_mono", 5: "traceback", 6: "emit"}


def build(force: bool = False) -> None:
    """compile liboracle.so (and oracle/_ref when /root/reference is present)"""
    if force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(LIB_PATH)
        for f in ("yama_oracle.c", "yama_profile_oracle.c", "ref_batch.c", "oracle.h")
    ):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and not os.path.exists(REF_PATH):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


class Scores(C.Structure):
    _fields_ = [("ss", (C.c_int * 128) * 128), ("gop", C.c_int * 16),
                ("gap_open", C.c_int), ("gap_extend", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        u8p, i32p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_int64)
        for name in ("mzo_yama_faithful", "mzo_yama_profile"):
            f = getattr(_lib, name)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                          C.c_void_p, C.c_void_p, C.POINTER(Scores), C.c_void_p,
                          C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        _lib.mzo_smooth.restype = None
        _lib.mzo_smooth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib.mzo_yama_check.restype = C.c_int
        _lib.mzo_yama_check.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        _lib.mzo_yama_batch.restype = C.c_int
        _lib.mzo_yama_batch.argtypes = [C.c_int] + [C.c_void_p] * 11 + [C.POINTER(Scores), C.c_int, C.c_int,
                                                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        _lib.mzo_scores_hoxd70.argtypes = [C.POINTER(Scores)]
        _lib.mzo_scores_hoxd85.argtypes = [C.POINTER(Scores)]
        del u8p, i32p, i64p
    return _lib


def scores70() -> Scores:
    s = Scores()
    lib().mzo_scores_hoxd70(C.byref(s))
    return s


def scores85() -> Scores:
    s = Scores()
    lib().mzo_scores_hoxd85(C.byref(s))
    return s


@dataclass
class YamaResult:
    rc: int
    OM: int = 0
    cols: Optional[np.ndarray] = None      # (OM, K+L) uint8
    final: Optional[np.ndarray] = None     # int32[3]  C,D,I at (M,N)
    tb: Optional[np.ndarray] = None        # band-packed traceback bytes (reference order)


def band_cells(LB, RB) -> int:
    LB = np.asarray(LB, dtype=np.int64)
    RB = np.asarray(RB, dtype=np.int64)
    return int((RB - LB + 1).sum())


def check(M: int, N: int, LB, RB) -> Tuple[int, int, int]:
    LB = np.ascontiguousarray(LB, dtype=np.int32)
    RB = np.ascontiguousarray(RB, dtype=np.int32)
    cells, bad = C.c_int64(0), C.c_int(-1)
    rc = lib().mzo_yama_check(M, N, LB.ctypes.data, RB.ctypes.data, C.byref(cells), C.byref(bad))
    return rc, cells.value, bad.value


def yama(A: np.ndarray, B: np.ndarray, LB, RB, sc: Optional[Scores] = None,
         variant: str = "faithful", want_tb: bool = False) -> YamaResult:
    """A: (M,K) uint8 column-major block (row index = column of the alignment), B: (N,L)."""
    A = np.ascontiguousarray(A, dtype=np.uint8)
    B = np.ascontiguousarray(B, dtype=np.uint8)
    M, K = A.shape
    N, L = B.shape
    LB = np.ascontiguousarray(LB, dtype=np.int32)
    RB = np.ascontiguousarray(RB, dtype=np.int32)
    assert LB.shape == (M + 1,) and RB.shape == (M + 1,)
    sc = sc or scores70()
    out = np.zeros(((M + N), K + L), dtype=np.uint8)
    om = C.c_int(0)
    final = np.zeros(3, dtype=np.int32)
    rc0, cells, _ = check(M, N, LB, RB)
    tb = np.zeros(max(cells, 1), dtype=np.uint8) if (want_tb and rc0 == 0) else None
    fn = lib().mzo_yama_faithful if variant == "faithful" else lib().mzo_yama_profile
    rc = fn(A.ctypes.data, K, M, B.ctypes.data, L, N, LB.ctypes.data, RB.ctypes.data,
            C.byref(sc), out.ctypes.data, C.byref(om), final.ctypes.data,
            tb.ctypes.data if tb is not None else None)
    if rc:
        return YamaResult(rc=rc)
    return YamaResult(rc=0, OM=om.value, cols=out[: om.value].copy(), final=final, tb=tb)


def smooth(LB, RB, M: int, N: int, radius: int):
    LB = np.ascontiguousarray(LB, dtype=np.int32).copy()
    RB = np.ascontiguousarray(RB, dtype=np.int32).copy()
    lib().mzo_smooth(LB.ctypes.data, RB.ctypes.data, M, N, radius)
    return LB, RB


def fnv1a(data: bytes, h: int = 0) -> int:
    if h == 0:
        h = 1469598103934665603
    for b in data:
        h ^= b
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def fnv1a_np(arr: np.ndarray, h: int = 0) -> int:
    """FNV-1a 64 over a contiguous byte array, in C (mzo_fnv1a)"""
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    f = lib().mzo_fnv1a
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_int64, C.c_uint64]
    return int(f(arr.ctypes.data, arr.size, h))


def _batch_args(batch: dict):
    arrs = [np.ascontiguousarray(batch[k], dtype=np.int32) for k in ("K", "L", "M", "N")]
    offs = [np.ascontiguousarray(batch[k], dtype=np.int64) for k in ("offA", "offB", "offBand")]
    pools = [np.ascontiguousarray(batch["poolA"], dtype=np.uint8), np.ascontiguousarray(batch["poolB"], dtype=np.uint8),
             np.ascontiguousarray(batch["poolLB"], dtype=np.int32), np.ascontiguousarray(batch["poolRB"], dtype=np.int32)]
    return arrs + offs + pools


def ref_batch(batch: dict, threads: int = 1, path: str = REF_PATH):
    """the compiled reference's own yama() over a packed batch (OpenMP, one pair per thread).
    Returns (om, hash, cells, n_rejected)."""
    n = len(batch["K"])
    om = np.zeros(n, dtype=np.int32)
    hs = np.zeros(n, dtype=np.uint64)
    cells = C.c_int64(0)
    keep = _batch_args(batch)
    f = lib().mzo_ref_batch
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_int] + [C.c_void_p] * 11 + [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    bad = f(path.encode(), n, *[a.ctypes.data for a in keep], threads, om.ctypes.data, hs.ctypes.data, C.byref(cells))
    if bad < 0:
        raise RuntimeError(f"cannot load the compiled reference {path}")
    return om, hs, cells.value, bad


def yama_batch(batch: dict, sc: Optional[Scores] = None, variant: int = 0, threads: int = 1):
    """batch: dict of packed pools as produced by multiz_amd.synth (K,L,M,N,offA,offB,offBand int arrays,
    poolA,poolB uint8, poolLB,poolRB int32).  Returns (om int32[n], hash uint64[n], cells, n_bad)."""
    n = len(batch["K"])
    sc = sc or scores70()
    om = np.zeros(n, dtype=np.int32)
    hs = np.zeros(n, dtype=np.uint64)
    cells = C.c_int64(0)
    arrs = [np.ascontiguousarray(batch[k], dtype=np.int32) for k in ("K", "L", "M", "N")]
    offs = [np.ascontiguousarray(batch[k], dtype=np.int64) for k in ("offA", "offB", "offBand")]
    pools = [np.ascontiguousarray(batch["poolA"], dtype=np.uint8), np.ascontiguousarray(batch["poolB"], dtype=np.uint8),
             np.ascontiguousarray(batch["poolLB"], dtype=np.int32), np.ascontiguousarray(batch["poolRB"], dtype=np.int32)]
    bad = lib().mzo_yama_batch(n, *[a.ctypes.data for a in arrs + offs + pools], C.byref(sc), variant, threads,
                               om.ctypes.data, hs.ctypes.data, C.byref(cells))
    return om, hs, cells.value, bad


# --------------------------------------------------------------------------------------
# The compiled reference (only where oracle/_ref/libref.so exists)
# --------------------------------------------------------------------------------------

class MafComp(C.Structure):
    pass


MafComp._fields_ = [  # reference maf.h:40-57; offsets verified in SURVEY.md section 8b
    ("next", C.POINTER(MafComp)), ("name", C.c_char_p), ("src", C.c_char_p), ("text", C.c_void_p),
    ("contig", C.c_char_p), ("mafPosMap", C.POINTER(C.c_int)), ("srcSize", C.c_int), ("start", C.c_int),
    ("size", C.c_int), ("nameID", C.c_short), ("strand", C.c_char), ("paralog", C.c_char)]


class MafAli(C.Structure):
    pass


MafAli._fields_ = [  # reference maf.h:29-37
    ("next", C.POINTER(MafAli)), ("score", C.c_double), ("components", C.POINTER(MafComp)),
    ("textSize", C.c_int), ("chain_len", C.c_int)]

assert C.sizeof(MafAli) == 32 and C.sizeof(MafComp) == 64


@dataclass
class Row:
    src: str
    start: int
    size: int
    strand: str
    srcSize: int
    text: str
    name: str = ""
    paralog: str = "o"


@dataclass
class Block:
    rows: List[Row] = field(default_factory=list)
    score: float = 0.0

    @property
    def textSize(self) -> int:
        return len(self.rows[0].text) if self.rows else 0


class _Keep:
    """keeps ctypes buffers alive for the lifetime of a marshalled mafAli"""
    def __init__(self):
        self.objs = []


def block_to_c(b: Block, keep: _Keep) -> MafAli:
    comps = []
    for r in b.rows:
        c = MafComp()
        tbuf = C.create_string_buffer(r.text.encode("ascii"))
        sbuf = C.create_string_buffer(r.src.encode("ascii"))
        nbuf = C.create_string_buffer((r.name or r.src.split(".")[0]).encode("ascii"))
        keep.objs += [tbuf, sbuf, nbuf, c]
        c.text = C.cast(tbuf, C.c_void_p)
        c.src = C.cast(sbuf, C.c_char_p)
        c.name = C.cast(nbuf, C.c_char_p)
        c.contig = None
        c.mafPosMap = None
        c.srcSize, c.start, c.size = r.srcSize, r.start, r.size
        c.nameID = 0
        c.strand = r.strand.encode("ascii")
        c.paralog = r.paralog.encode("ascii")
        comps.append(c)
    for i in range(len(comps) - 1):
        comps[i].next = C.pointer(comps[i + 1])
    a = MafAli()
    a.next = None
    a.score = b.score
    a.components = C.pointer(comps[0]) if comps else None
    a.textSize = b.textSize
    a.chain_len = 0
    keep.objs.append(a)
    return a


def block_from_c(p) -> Optional[Block]:
    if not p:
        return None
    a = p.contents
    rows = []
    cp = a.components
    while cp:
        c = cp.contents
        text = C.string_at(c.text).decode("ascii")
        rows.append(Row(src=c.src.decode("ascii"), start=c.start, size=c.size, strand=c.strand.decode("ascii"),
                        srcSize=c.srcSize, text=text, name=(c.name or b"").decode("ascii"),
                        paralog=(c.paralog or b"o").decode("ascii") if c.paralog != b"\x00" else "o"))
        cp = c.next
    return Block(rows=rows, score=a.score)


class Reference:
    """oracle/_ref/libref.so -- the reference's own yama()/smooth()/pre_yama(), unmodified."""

    def __init__(self, path: str = REF_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.lib.init_scores70()
        self.lib.yama.restype = None
        self.lib.pre_yama.restype = C.POINTER(MafAli)
        self.lib.pre_yama.argtypes = [C.POINTER(MafAli), C.POINTER(MafAli), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        self.lib.mafScoreRange.restype = C.c_double
        self.lib.mafScoreRange.argtypes = [C.POINTER(MafAli), C.c_int, C.c_int]
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]

    def scores(self, which: int = 70):
        (self.lib.init_scores70 if which == 70 else self.lib.init_scores85)()

    @staticmethod
    def _cols_1based(X: np.ndarray):
        n = X.shape[0]
        ptrs = (C.c_void_p * (n + 1))()
        base = X.ctypes.data
        stride = X.shape[1]
        for i in range(1, n + 1):
            ptrs[i] = base + (i - 1) * stride
        return ptrs

    def smooth(self, LB, RB, M, N, radius):
        LB = np.ascontiguousarray(LB, dtype=np.int32).copy()
        RB = np.ascontiguousarray(RB, dtype=np.int32).copy()
        self.lib.smooth(C.c_void_p(LB.ctypes.data), C.c_void_p(RB.ctypes.data), C.c_int(M), C.c_int(N), C.c_int(radius))
        return LB, RB

    def yama(self, A: np.ndarray, B: np.ndarray, LB, RB) -> YamaResult:
        """inputs must satisfy the validity prologue (the reference exit(1)s otherwise)"""
        A = np.ascontiguousarray(A, dtype=np.uint8)
        B = np.ascontiguousarray(B, dtype=np.uint8)
        M, K = A.shape
        N, L = B.shape
        LB = np.ascontiguousarray(LB, dtype=np.int32)
        RB = np.ascontiguousarray(RB, dtype=np.int32)
        pa, pb = self._cols_1based(A), self._cols_1based(B)
        oal = C.POINTER(C.c_void_p)()
        om = C.c_int(0)
        self.lib.yama(pa, C.c_int(K), C.c_int(M), pb, C.c_int(L), C.c_int(N),
                      C.c_void_p(LB.ctypes.data), C.c_void_p(RB.ctypes.data), C.byref(oal), C.byref(om))
        n = om.value
        first = oal[1]
        cols = np.frombuffer(C.string_at(first, n * (K + L)), dtype=np.uint8).reshape(n, K + L).copy()
        self.libc.free(C.c_void_p(first))
        self.libc.free(C.c_void_p(C.addressof(oal.contents) + C.sizeof(C.c_void_p)))
        return YamaResult(rc=0, OM=n, cols=cols)

    def pre_yama(self, a1: Block, a2: Block, beg: int, end: int, radius: int, v: int) -> Optional[Block]:
        keep = _Keep()
        c1, c2 = block_to_c(a1, keep), block_to_c(a2, keep)
        p = self.lib.pre_yama(C.byref(c1), C.byref(c2), beg, end, radius, v, None)
        return block_from_c(p)

    def score_range(self, b: Block, start: int, size: int) -> float:
        keep = _Keep()
        c = block_to_c(b, keep)
        return self.lib.mafScoreRange(C.byref(c), start, size)


def have_reference() -> bool:
    return os.path.exists(REF_PATH)
