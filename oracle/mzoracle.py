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


def hash_cols(cols_ptr: np.ndarray, om: np.ndarray, width: np.ndarray) -> np.ndarray:
    """per-pair FNV-1a of (OM, merged columns) for columns given by ADDRESS (uint64; 0 = none) -- the same hash as
    ref_batch() / yama_batch() return, over the product's mz_out.cols of a host-path call"""
    n = len(om)
    ptr = np.ascontiguousarray(cols_ptr, dtype=np.uint64)
    om = np.ascontiguousarray(om, dtype=np.int32)
    width = np.ascontiguousarray(width, dtype=np.int32)
    out = np.zeros(n, dtype=np.uint64)
    f = lib().mzo_hash_cols
    f.restype = None
    f.argtypes = [C.c_int] + [C.c_void_p] * 4
    f(n, ptr.ctypes.data, om.ctypes.data, width.ctypes.data, out.ctypes.data)
    return out


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
    paralog: str = "s"          # 's'ingleton, the reader's default (maf.c:178)


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
        _nm, _, _rest = r.src.partition(".")
        cbuf = C.create_string_buffer((_rest or _nm).encode("ascii"))      # parseSrcName2 (multi_util.c:909-925)
        keep.objs.append(cbuf)
        c.contig = C.cast(cbuf, C.c_char_p)
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
                        paralog=c.paralog.decode("ascii") if c.paralog not in (b"", b"\x00") else "s"))
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


# --------------------------------------------------------------------------------------
# pre_yama() restated in Python (reference mz_preyama.c:17-359, mz_scores.c:124-152,
# multi_util.c:570-645).  Small blocks only; the DP itself goes to the C oracle.
# --------------------------------------------------------------------------------------

DASH = ord("-")


def _ss_table(sc: Scores) -> np.ndarray:
    return np.ctypeslib.as_array(sc.ss).reshape(128, 128).copy()


def score_range(b: Block, start: int, size: int, sc: Optional[Scores] = None) -> float:
    """mafScoreRange (mz_scores.c:124-152): all unordered row pairs, substitution minus gap-open"""
    sc = sc or scores70()
    ss = _ss_table(sc)
    gop = np.array(list(sc.gop), dtype=np.int64)
    rows = [np.frombuffer(r.text.encode("ascii"), dtype=np.uint8) for r in b.rows]
    total = 0
    for p in range(len(rows)):
        for q in range(p + 1, len(rows)):
            x, y = rows[p][start:start + size].astype(np.int64), rows[q][start:start + size].astype(np.int64)
            total += int(ss[x, y].sum())
            lo = max(start, 1)
            if start + size > lo:
                cx, cy = rows[p][lo:start + size] == DASH, rows[q][lo:start + size] == DASH
                px, py = rows[p][lo - 1:start + size - 1] == DASH, rows[q][lo - 1:start + size - 1] == DASH
                idx = (px.astype(np.int64) << 3) | (py.astype(np.int64) << 2) | (cx.astype(np.int64) << 1) | cy.astype(np.int64)
                total -= int(gop[idx].sum())
    return float(total)


def pos2col(r: Row, pos: int, text_size: int) -> int:
    if pos < r.start or pos >= r.start + r.size:
        raise ValueError(f"mafPos2Col: {pos} not in {r.start}-{r.start + r.size - 1}")
    p = r.start - 1
    for col in range(text_size):
        if r.text[col] != "-":
            p += 1
            if p == pos:
                return col
    return text_size


class _Cols:
    """1-based columns over one flat buffer (+ one spare non-dash byte), as the reference lays them out"""

    def __init__(self, ncols: int, nrows: int):
        self.rows, self.n = nrows, ncols
        self.buf = bytearray(ncols * nrows + 1)
        self.buf[ncols * nrows] = ord("N")

    def at(self, col: int, row: int) -> int:          # row may run past nrows-1: next column's bytes
        return self.buf[(col - 1) * self.rows + row]

    def set(self, col, row, v):
        self.buf[(col - 1) * self.rows + row] = v

    def col(self, c) -> bytes:
        return bytes(self.buf[(c - 1) * self.rows: c * self.rows])

    def as_array(self, ncols) -> np.ndarray:
        return np.frombuffer(bytes(self.buf[: ncols * self.rows]), dtype=np.uint8).reshape(ncols, self.rows)


def _rm_col_dash(X: _Cols, n: int):
    """rmColDash (mz_preyama.c:87-108): compact in place, return (map[0..n], new n)"""
    mp = [-1] * (n + 1)
    kept = 0
    for i in range(1, n + 1):
        if all(X.at(i, j) == DASH for j in range(X.rows)):
            continue
        kept += 1
        if kept != i:
            for j in range(X.rows):
                X.set(kept, j, X.at(i, j))
        mp[i] = kept
    return mp, kept


def _mapping(Aat, a_r1, a_r2, a_c1, a_c2, Bat, b_r1, b_r2, b_c1, b_c2):
    """mapping (mz_preyama.c:111-148); Aat/Bat are (col,row)->byte accessors"""
    mp = {i: -1 for i in range(a_c1, a_c2 + 1)}
    i, k = a_c1, b_c1
    while i <= a_c2 and k <= b_c2:
        a_live = b_live = False
        while i <= a_c2:
            if any(Aat(i, j) != DASH for j in range(a_r1, a_r2 + 1)):
                a_live = True
                break
            i += 1
        while k <= b_c2:
            if any(Bat(k, l) != DASH for l in range(b_r1, b_r2 + 1)):
                b_live = True
                break
            k += 1
        if a_live and b_live:
            mp[i] = k
        i += 1
        k += 1
    return mp


def _note(LB, RB, row, col, unset_hi):
    if LB[row] == 0 or LB[row] > col:
        LB[row] = col
    if RB[row] == unset_hi or RB[row] < col:
        RB[row] = col


def _build(cols: np.ndarray, a1: Block, cbeg1: int, a2: Block, cbeg2: int, sc) -> Optional[Block]:
    """mafBuild with top == 0 (mz_preyama.c:38-81)"""
    ncol, nrow = cols.shape
    srcs = [(r, cbeg1) for r in a1.rows] + [(r, cbeg2) for r in a2.rows[1:]]
    out = []
    for i in range(nrow):
        src, skip = srcs[i]
        text = bytes(cols[:, i]).decode("ascii")
        size = sum(ch != "-" for ch in text)
        if size == 0:
            continue
        start = src.start + sum(ch != "-" for ch in src.text[:skip])
        out.append(Row(src=src.src, start=start, size=size, strand=src.strand, srcSize=src.srcSize, text=text,
                       name=src.name, paralog=src.paralog))
    if not out:
        return None
    b = Block(rows=out)
    b.score = score_range(b, 0, ncol, sc)
    return b


def part_ali_col(b: Block, cbeg: int, cend: int, sc=None) -> Optional[Block]:
    """make_part_ali_col (multi_util.c:570-618)"""
    if cend - cbeg + 1 == 0:
        return None
    rows = []
    for r in b.rows:
        seg = r.text[cbeg:cend + 1]
        bases = sum(ch != "-" for ch in seg)
        if bases == 0:
            continue
        rows.append(Row(src=r.src, start=r.start + sum(ch != "-" for ch in r.text[:cbeg]), size=bases, strand=r.strand,
                        srcSize=r.srcSize, text=seg, name=r.name, paralog=r.paralog))
    if not rows:
        return None
    keep = [j for j in range(cend - cbeg + 1) if any(r.text[j] != "-" for r in rows)]
    for r in rows:
        r.text = "".join(r.text[j] for j in keep)
    out = Block(rows=rows)
    out.score = score_range(out, 0, len(keep), sc)
    return out


def pre_yama(a1: Block, a2: Block, beg: int, end: int, radius: int, v: int, sc: Optional[Scores] = None,
             yama_fn=None):
    """Returns (merged block or None, side block or None); the side block is what the reference
    writes to fpw2 when v == 0 and a1 has nothing but its reference row (mz_preyama.c:193-201).
    yama_fn(A, B, LB, RB) -> YamaResult lets a test substitute the implementation under test."""
    sc = sc or scores70()
    yfn = yama_fn or (lambda A, B, LB, RB: yama(A, B, LB, RB, sc=sc))
    K, L = len(a1.rows), len(a2.rows) - 1
    ts1, ts2 = a1.textSize, a2.textSize
    cbeg1, cend1 = pos2col(a1.rows[0], beg, ts1), pos2col(a1.rows[0], end, ts1)
    cbeg2, cend2 = pos2col(a2.rows[0], beg, ts2), pos2col(a2.rows[0], end, ts2)
    M = M_all = cend1 - cbeg1 + 1
    N = N_all = cend2 - cbeg2 + 1

    B = _Cols(N, L)
    for i in range(1, N + 1):
        for r in range(L):
            B.set(i, r, ord(a2.rows[r + 1].text[cbeg2 + i - 1]))
    map2, N = _rm_col_dash(B, N)
    if N < 1:
        return None, None
    if v == 0:
        K -= 1
    if K == 0:
        return None, part_ali_col(a2, cbeg2, cend2, sc)
    first = 1 if v == 0 else 0
    A = _Cols(M, K)
    for i in range(1, M + 1):
        for r in range(K):
            A.set(i, r, ord(a1.rows[first + r].text[cbeg1 + i - 1]))
    if v == 0:
        map1, M = _rm_col_dash(A, M)
        if M < 1:
            return None, None
    else:
        map1 = list(range(M + 1))

    LB, RB = [0] * (M + 1), [N] * (M + 1)
    t1, t2 = a1.rows[0].text, a2.rows[0].text
    i, j = cbeg1, cbeg2
    while i <= cend1:
        while t1[i] == "-":
            i += 1
        while t2[j] == "-":
            j += 1
        ra, cb = map1[i - cbeg1 + 1], map2[j - cbeg2 + 1]
        if ra != -1 and cb != -1:
            _note(LB, RB, ra, cb, N)
        i += 1
        j += 1
    LB, RB = smooth(LB, RB, M, N, radius)
    res = yfn(A.as_array(M), B.as_array(N), LB, RB)
    if res.rc:
        raise RuntimeError(f"yama rejected the band: {ERRORS.get(res.rc, res.rc)}")
    merged, M_new = res.cols, res.OM
    if v == 1:
        return _build(merged, a1, cbeg1, a2, cbeg2, sc), None

    # ---- v == 0: align a1's reference row against the merged block
    ref1 = _Cols(M_all, 1)
    for i in range(1, M_all + 1):
        ref1.set(i, 0, ord(t1[cbeg1 + i - 1]))
    m3a, M3 = _rm_col_dash(ref1, M_all)
    mat = lambda c, r: int(merged[c - 1, r])          # noqa: E731
    # rows 1..K of a K-row matrix: the reference's off-by-one (mz_preyama.c:279), reproduced
    m4a = _mapping(A.at, 1, K, 1, M, mat, 0, K - 1, 1, M_new)
    LBa, RBa = [0] * (M3 + 1), [M_new] * (M3 + 1)
    for i in range(1, M_all + 1):
        if map1[i] == -1:
            continue
        a, b_ = m3a[i], m4a[map1[i]]
        if a != -1 and b_ != -1:
            _note(LBa, RBa, a, b_, M_new)
    LBa, RBa = smooth(LBa, RBa, M3, M_new, radius)

    ref2 = _Cols(N_all, 1)
    for i in range(1, N_all + 1):
        ref2.set(i, 0, ord(t2[cbeg2 + i - 1]))
    m3b, N3 = _rm_col_dash(ref2, N_all)
    m4b = _mapping(B.at, 0, L - 1, 1, N, mat, K, K + L - 1, 1, M_new)
    LBb, RBb = [0] * (N3 + 1), [M_new] * (N3 + 1)
    for i in range(1, N_all + 1):
        a = m3b[i]
        b_ = 0 if map2[i] == -1 else m4b[map2[i]]   # map4[-1] reads 0 on glibc (mz_preyama.c:318-326)
        if a != -1 and b_ != -1:
            _note(LBb, RBb, a, b_, M_new)
    LBb, RBb = smooth(LBb, RBb, N3, M_new, radius)
    if M3 != N3:
        raise RuntimeError("M3 not equals N3!!")
    LBf = np.minimum(LBa, LBb)
    RBf = np.maximum(RBa, RBb)
    res2 = yfn(ref1.as_array(M3), merged, LBf, RBf)
    if res2.rc:
        raise RuntimeError(f"second yama rejected the band: {ERRORS.get(res2.rc, res2.rc)}")
    return _build(res2.cols, a1, cbeg1, a2, cbeg2, sc), None


def format_block(b: Block) -> str:
    """mafWrite (maf.c:251-294)"""
    out = ["a" + ("" if b.score == float(-(1 << 31)) else f" score={b.score:3.1f}")]
    w = [max(len(r.src) for r in b.rows), max(len(str(r.start)) for r in b.rows),
         max(len(str(r.size)) for r in b.rows), max(len(str(r.srcSize)) for r in b.rows)]
    for r in b.rows:
        name, _, rest = r.src.partition(".")
        shown = name if (not rest or rest == name) else f"{name}.{rest}"
        out.append(f"s {shown:<{w[0]}} {r.start:>{w[1]}} {r.size:>{w[2]}} {r.strand} {r.srcSize:>{w[3]}} {r.text}")
    return "\n".join(out) + "\n\n"
