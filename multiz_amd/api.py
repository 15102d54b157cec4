"""ctypes binding of libmzamd.so (include/mz_amd.h).  No fallback of any kind: if the library
is missing or no HIP device is usable, calls raise."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")      # before the HIP runtime starts: see mz_host.c, init_devices()

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MZ_LIB_PATH") or os.path.join(HERE, "libmzamd.so")     # (MZ_LIB_PATH: experimental builds, tests/tools)
CSRC = os.path.join(HERE, "csrc")

MZ_STATUS = {0: "ok", 1: "termination", 2: "narrow", 3: "lb_mono", 4: "rb_mono", 5: "traceback", 6: "emit",
             16: "rows", 17: "shape", 18: "range", 19: "workspace", 20: "device", 21: "sentinel"}


def build(force: bool = False) -> str:
    """compile libmzamd.so for gfx950 in-tree (hipcc + gcc via multiz_amd/csrc/Makefile)"""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


class Job(C.Structure):
    _fields_ = [("K", C.c_int), ("L", C.c_int), ("M", C.c_int), ("N", C.c_int),
                ("A", C.c_void_p), ("B", C.c_void_p), ("LB", C.c_void_p), ("RB", C.c_void_p)]


class Out(C.Structure):
    _fields_ = [("status", C.c_int), ("badrow", C.c_int), ("OM", C.c_int), ("score", C.c_int * 3),
                ("cols", C.c_void_p), ("block", C.c_void_p)]


class DevBatchC(C.Structure):
    _fields_ = [("n", C.c_int32), ("dp_hint", C.c_int32), ("dp_grid", C.c_int32), ("dp_rows", C.c_int32), ("hint_gen", C.c_int32), ("walk_hint", C.c_int32)] + \
        [(k, C.c_void_p) for k in ("K", "L", "M", "N", "offA", "offB", "offBand", "poolA", "poolB", "poolLB", "poolRB",
                                   "status", "badrow", "mode", "cells", "edgeLo", "edgeHi", "szTb", "szScript", "szOut", "szPrep",
                                   "offTb", "offScript", "offOut", "offPrep", "totals", "packList", "scanAux",
                                   "tbw", "script", "out", "prep")] + \
        [("capTb", C.c_int64), ("capScript", C.c_int64), ("capOut", C.c_int64), ("capPrep", C.c_int64),
         ("om", C.c_void_p), ("final3", C.c_void_p)]


_lib = None
MZ_AMD_ABI = 3          # include/mz_amd.h


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing -- run __graft_entry__.build() (there is no fallback path)")
        if os.environ.get("MZ_NO_TORCH") != "1":       # (host-only helpers in a process that never touches the GPU)
            try:
                # torch ships its own libamdhip64; load it first so that this library binds to the same
                # HIP runtime instead of bringing /opt/rocm's copy into the process as a second one
                import torch  # noqa: F401
            except ImportError:
                pass
        _lib = C.CDLL(LIB_PATH)
        got = _lib.mz_abi_version() if hasattr(_lib, "mz_abi_version") else 0
        if got != MZ_AMD_ABI:                           # (the structures below are laid out for exactly this revision)
            _lib = None
            raise RuntimeError(f"{LIB_PATH} has ABI {got}, these bindings are for {MZ_AMD_ABI}: rebuild (__graft_entry__.build())")
        _lib.mz_last_error.restype = C.c_char_p
        _lib.mz_init.argtypes = [C.c_int]
        _lib.mz_stream.restype = C.c_void_p
        _lib.mz_yama_batch.argtypes = [C.c_int, C.POINTER(Job), C.POINTER(Out)]
        _lib.mz_set_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib.mz_dev_plan_bytes.restype = C.c_size_t
        _lib.mz_dev_plan_bytes.argtypes = [C.c_int]
        _lib.mz_dev_carve.argtypes = [C.POINTER(DevBatchC), C.c_void_p]
        for f in ("mz_dev_plan", "mz_dev_dp", "mz_dev_walk", "mz_dev_emit"):
            getattr(_lib, f).argtypes = [C.POINTER(DevBatchC), C.c_void_p]
        _lib.mz_dev_run.argtypes = [C.POINTER(DevBatchC), C.c_void_p, C.POINTER(C.c_float)]
        _lib.mz_dev_run_async.argtypes = [C.POINTER(DevBatchC), C.c_void_p, C.c_void_p]
        _lib.mz_dev_wait.argtypes = [C.c_void_p]
        _lib.free_cols = C.CDLL(None).free
        _lib.free_cols.argtypes = [C.c_void_p]
    return _lib


def _check(rc: int, what: str):
    if rc < 0:
        raise RuntimeError(f"{what}: {lib().mz_last_error().decode()}")


def init(device: int = 0):
    _check(lib().mz_init(device), "mz_init")


def init_multi(ngpu: int, devices: Optional[Sequence[int]] = None):
    """several GPUs in ONE process (mz_init_multi): mz_yama_batch() then deals a host list out over them"""
    f = lib().mz_init_multi
    f.argtypes = [C.c_int, C.POINTER(C.c_int)]
    arr = (C.c_int * ngpu)(*devices) if devices is not None else None
    _check(f(ngpu, arr), "mz_init_multi")


def device_identity(ctx: int = 0) -> str:
    """'<PCI bus id> <device name>' of the GPU a context runs on"""
    buf = C.create_string_buffer(256)
    f = lib().mz_device_identity
    f.argtypes = [C.c_int, C.c_char_p, C.c_int]
    return buf.value.decode() if f(ctx, buf, 256) == 0 else "?"


def set_scores_hoxd70():
    lib().init_scores70()


def set_scores_hoxd85():
    lib().init_scores85()


class Result:
    __slots__ = ("status", "badrow", "OM", "score", "cols")

    def __init__(self, status, badrow, OM, score, cols):
        self.status, self.badrow, self.OM, self.score, self.cols = status, badrow, OM, score, cols


def yama_batch(pairs: Sequence[tuple]) -> List[Result]:
    """pairs: sequence of (A (M,K) uint8, B (N,L) uint8, LB int32[M+1], RB int32[M+1]).
    Runs them as one GPU batch through the C ABI (mz_yama_batch)."""
    n = len(pairs)
    jobs = (Job * n)()
    outs = (Out * n)()
    keep = []
    for i, (A, B, LB, RB) in enumerate(pairs):
        A = np.ascontiguousarray(A, dtype=np.uint8)
        B = np.ascontiguousarray(B, dtype=np.uint8)
        LB = np.ascontiguousarray(LB, dtype=np.int32)
        RB = np.ascontiguousarray(RB, dtype=np.int32)
        keep += [A, B, LB, RB]
        jobs[i].M, jobs[i].K = A.shape
        jobs[i].N, jobs[i].L = B.shape
        jobs[i].A, jobs[i].B, jobs[i].LB, jobs[i].RB = A.ctypes.data, B.ctypes.data, LB.ctypes.data, RB.ctypes.data
    f = lib().mz_free_outs
    f.argtypes = [C.c_int, C.c_void_p]
    rc = lib().mz_yama_batch(n, jobs, outs)
    try:
        _check(rc, "mz_yama_batch")                      # (an error return leaves the blocks of the chunks that did come back: freed below)
        res = []
        for i in range(n):
            o = outs[i]
            cols = None
            if o.status == 0:
                w = jobs[i].K + jobs[i].L
                cols = np.frombuffer(C.string_at(o.cols, o.OM * w), dtype=np.uint8).reshape(o.OM, w).copy()
            res.append(Result(o.status, o.badrow, o.OM, np.array(list(o.score), dtype=np.int32), cols))
    finally:
        f(n, C.cast(outs, C.c_void_p))                   # the call's result blocks (one per chunk: mz_out.block)
    return res


JOB_DT = np.dtype([("K", "<i4"), ("L", "<i4"), ("M", "<i4"), ("N", "<i4"), ("A", "<u8"), ("B", "<u8"), ("LB", "<u8"), ("RB", "<u8")])
OUT_DT = np.dtype([("status", "<i4"), ("badrow", "<i4"), ("OM", "<i4"), ("score", "<i4", (3,)), ("cols", "<u8"), ("block", "<u8")])


def host_jobs(batch: dict):
    """mz_job[n] / mz_out[n] (include/mz_amd.h) as numpy record arrays over a packed host batch, without a per-pair
    loop: what a C caller of mz_yama_batch() would hold.  The batch's arrays must stay alive while the jobs are used."""
    assert JOB_DT.itemsize == C.sizeof(Job) and OUT_DT.itemsize == C.sizeof(Out)
    n = len(batch["K"])
    for k, dt in (("poolA", np.uint8), ("poolB", np.uint8), ("poolLB", np.int32), ("poolRB", np.int32)):
        assert batch[k].dtype == dt and batch[k].flags["C_CONTIGUOUS"], k
    jobs = np.zeros(n, dtype=JOB_DT)
    for k in ("K", "L", "M", "N"):
        jobs[k] = batch[k]
    jobs["A"] = batch["poolA"].ctypes.data + batch["offA"].astype(np.uint64)
    jobs["B"] = batch["poolB"].ctypes.data + batch["offB"].astype(np.uint64)
    jobs["LB"] = batch["poolLB"].ctypes.data + 4 * batch["offBand"].astype(np.uint64)
    jobs["RB"] = batch["poolRB"].ctypes.data + 4 * batch["offBand"].astype(np.uint64)
    return jobs, np.zeros(n, dtype=OUT_DT)


def yama_batch_records(jobs: np.ndarray, outs: np.ndarray) -> int:
    """mz_yama_batch() on record arrays from host_jobs(); the caller reads outs and then calls free_outs()"""
    rc = lib().mz_yama_batch(len(jobs), jobs.ctypes.data_as(C.POINTER(Job)), outs.ctypes.data_as(C.POINTER(Out)))
    _check(rc, "mz_yama_batch")
    return rc


def link_bytes(jobs: np.ndarray):
    """(bytes to the device, bytes back) of one mz_yama_batch() call over these jobs: the library's own accounting
    (mz_link_bytes: staging block sizes of the last call)"""
    up, down = C.c_int64(0), C.c_int64(0)
    lib().mz_link_bytes(C.byref(up), C.byref(down))
    return up.value, down.value


def free_outs(outs: np.ndarray):
    f = lib().mz_free_outs
    f.argtypes = [C.c_int, C.c_void_p]
    f(len(outs), outs.ctypes.data)


# ---------------------------------------------------------------- link images (include/mz_amd.h: mz_link_*)
class LinkDesc(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n", "image_bytes", "exc_bytes", "colsA", "colsB", "band", "steps", "res_bytes")]


RES_DT = np.dtype([("status", "<i4"), ("badrow", "<i4"), ("om", "<i4"), ("f", "<i4", (3,)), ("off", "<i8"), ("cells", "<i8")])    # mz_res_rec
LINK_PARTS = ("K", "L", "M", "N", "offA", "offB", "offBand", "len", "lb0", "rb0", "offC", "fmt", "steps", "nibA", "nibB", "bytes")


def _al256(x: int) -> int:
    return (x + 255) & ~255


def link_pack(jobs: np.ndarray):
    """mz_link_pack(): the jobs (records from host_jobs(), any subset in any order) as a link image.  Host only.
    Returns (desc int64[8], image uint8[...], exceptions uint8[...]) -- numpy copies, the library's buffers are released."""
    L = lib()
    d, img, exc = LinkDesc(), C.c_void_p(), C.c_void_p()
    jobs = np.ascontiguousarray(jobs)
    L.mz_link_pack.argtypes = [C.c_int, C.c_void_p, C.POINTER(LinkDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.mz_link_free.argtypes = [C.c_void_p]
    _check(L.mz_link_pack(len(jobs), jobs.ctypes.data, C.byref(d), C.byref(img), C.byref(exc)), "mz_link_pack")
    try:
        image = np.ctypeslib.as_array((C.c_uint8 * d.image_bytes).from_address(img.value)).copy() if d.image_bytes else np.zeros(0, np.uint8)
        excs = np.ctypeslib.as_array((C.c_uint8 * d.exc_bytes).from_address(exc.value)).copy() if d.exc_bytes else np.zeros(0, np.uint8)
    finally:
        L.mz_link_free(img)
        L.mz_link_free(exc)
    return np.array([getattr(d, k) for k, _ in LinkDesc._fields_], dtype=np.int64), image, excs


def _desc(desc: np.ndarray) -> LinkDesc:
    d = LinkDesc()
    for (k, _), v in zip(LinkDesc._fields_, np.asarray(desc, dtype=np.int64).tolist()):
        setattr(d, k, int(v))
    return d


def link_parts(desc: np.ndarray) -> dict:
    """byte offsets of the image's parts (mz_link_parts)"""
    at = (C.c_int64 * 16)()
    d = _desc(desc)
    if lib().mz_link_parts(C.byref(d), at) != 0:
        raise RuntimeError("mz_link_parts: " + lib().mz_last_error().decode())
    return dict(zip(LINK_PARTS, list(at)))


def link_run(desc: np.ndarray, image, exc):
    """The rank that aligns: image / exc are uint8 tensors ON THE GPU (as they arrived over RCCL).  Plans and runs the image on
    torch's current stream and returns the result image as a device tensor (records + 2-bit scripts: what goes back)."""
    import torch
    if image.device.type != "cuda":
        raise RuntimeError("link_run needs HIP device tensors (there is no CPU path in the product)")
    torch.cuda.set_device(image.device)
    L = lib()
    d = _desc(desc)
    st = torch.cuda.current_stream(image.device).cuda_stream
    L.mz_link_plan.argtypes = [C.POINTER(LinkDesc), C.c_void_p, C.c_void_p, C.c_void_p]
    L.mz_link_finish.argtypes = [C.POINTER(LinkDesc), C.c_void_p, C.c_void_p]
    _check(L.mz_link_plan(C.byref(d), image.data_ptr() if image.numel() else None, exc.data_ptr() if exc.numel() else None, st), "mz_link_plan")
    res = torch.zeros(int(d.res_bytes), dtype=torch.uint8, device=image.device)
    _check(L.mz_link_finish(C.byref(d), res.data_ptr(), st), "mz_link_finish")
    return res


def link_records(result, n: int):
    """the n records of a result image (numpy uint8 array or torch tensor on any device): a structured view / copy"""
    raw = result[64: 64 + RES_DT.itemsize * n]
    if not isinstance(raw, np.ndarray):
        raw = raw.cpu().numpy()
    return np.ascontiguousarray(raw).view(RES_DT)


def link_assemble(jobs: np.ndarray, result: np.ndarray) -> np.ndarray:
    """mz_link_assemble(): outs (OUT_DT records) for these jobs from their result image; release with free_outs().  Host only."""
    jobs = np.ascontiguousarray(jobs)
    result = np.ascontiguousarray(result, dtype=np.uint8)
    outs = np.zeros(len(jobs), dtype=OUT_DT)
    f = lib().mz_link_assemble
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    _check(f(len(jobs), jobs.ctypes.data, result.ctypes.data, result.size, outs.ctypes.data), "mz_link_assemble")
    return outs


# ---------------------------------------------------------------------------- include/mz_shard.h: the exchange in C
# (multiz_amd/csrc/mz_shard.c: scatter / align / gather of a list that exists on one rank, over a table of transport functions)

_SEND_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
_GROUP_T = C.CFUNCTYPE(C.c_int, C.c_void_p)


class Comm:
    """an mz_comm of the library: RCCL (`Comm.rccl`), every rank in this process (`Comm.loopback`), or the caller's own blocking
    send / recv of host bytes (`Comm.custom`: send(buf: np.ndarray uint8, peer), recv(buf, peer))"""

    def __init__(self, ptr, rank, size, keep=()):
        self.ptr, self.rank, self.size, self._keep = ptr, rank, size, keep

    @staticmethod
    def loopback(size: int):
        arr = (C.c_void_p * size)()
        f = lib().mz_comm_loopback
        f.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        _check(f(size, arr), "mz_comm_loopback")
        return [Comm(arr[r], r, size) for r in range(size)]

    @staticmethod
    def rccl_unique_id() -> bytes:
        buf = (C.c_char * 128)()
        _check(lib().mz_comm_rccl_unique_id(buf), "mz_comm_rccl_unique_id")
        return bytes(buf)

    @staticmethod
    def rccl(unique_id: bytes, rank: int, size: int):
        p = C.c_void_p()
        f = lib().mz_comm_rccl_create
        f.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        _check(f(bytes(unique_id), rank, size, C.byref(p)), "mz_comm_rccl_create")
        return Comm(p.value, rank, size)

    @staticmethod
    def custom(rank: int, size: int, send, recv):
        def view(buf, n):
            return np.ctypeslib.as_array((C.c_uint8 * n).from_address(buf)) if n else np.zeros(0, np.uint8)

        def c_send(_u, buf, n, peer):
            try:
                send(view(buf, n), peer)
                return 0
            except Exception:                              # (into the library's error return)
                return -1

        def c_recv(_u, buf, n, peer):
            try:
                recv(view(buf, n), peer)
                return 0
            except Exception:
                return -1
        cs, cr = _SEND_T(c_send), _SEND_T(c_recv)
        p = C.c_void_p()
        f = lib().mz_comm_custom
        f.argtypes = [C.c_int, C.c_int, C.c_void_p, _SEND_T, _SEND_T, _GROUP_T, _GROUP_T, C.POINTER(C.c_void_p)]
        _check(f(rank, size, None, cs, cr, C.cast(None, _GROUP_T), C.cast(None, _GROUP_T), C.byref(p)), "mz_comm_custom")
        return Comm(p.value, rank, size, keep=(cs, cr))

    def echo(self, nbytes: int):
        f = lib().mz_comm_echo
        f.argtypes = [C.c_void_p, C.c_size_t]
        _check(f(self.ptr, nbytes), "mz_comm_echo")

    def free(self):
        if self.ptr:
            f = lib().mz_comm_free
            f.argtypes = [C.c_void_p]
            f(self.ptr)
            self.ptr = None


class Shard:
    """one rank's share of a scattered list (mz_shard)"""

    def __init__(self, comm: Comm, root: int, jobs: Optional[np.ndarray]):
        L = lib()
        self.comm, self.root = comm, root
        self.jobs = np.ascontiguousarray(jobs) if jobs is not None else None
        p = C.c_void_p()
        L.mz_shard_scatter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        _check(L.mz_shard_scatter(comm.ptr, root, len(self.jobs) if self.jobs is not None else 0,
                                  self.jobs.ctypes.data if self.jobs is not None and len(self.jobs) else None, C.byref(p)), "mz_shard_scatter")
        self.ptr = p.value
        L.mz_shard_desc.restype = C.POINTER(LinkDesc)
        L.mz_shard_desc.argtypes = [C.c_void_p]
        d = L.mz_shard_desc(self.ptr).contents
        self.desc = np.array([getattr(d, k) for k, _ in LinkDesc._fields_], dtype=np.int64)
        self.n = int(d.n)
        L.mz_shard_index.restype = C.POINTER(C.c_int64)
        L.mz_shard_index.argtypes = [C.c_void_p]
        self.index = np.ctypeslib.as_array(L.mz_shard_index(self.ptr), shape=(self.n,)).copy() if self.n else np.zeros(0, np.int64)

    def host_image(self):
        """(image, exceptions) as numpy copies: for callers that align the share themselves"""
        img, exc = C.c_void_p(), C.c_void_p()
        f = lib().mz_shard_host_image
        f.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        _check(f(self.comm.ptr, self.ptr, C.byref(img), C.byref(exc)), "mz_shard_host_image")
        nb, ne = int(self.desc[1]), int(self.desc[2])
        image = np.ctypeslib.as_array((C.c_uint8 * nb).from_address(img.value)).copy() if nb else np.zeros(0, np.uint8)
        excs = np.ctypeslib.as_array((C.c_uint8 * ne).from_address(exc.value)).copy() if ne else np.zeros(0, np.uint8)
        return image, excs

    def set_result(self, result: np.ndarray):
        result = np.ascontiguousarray(result, dtype=np.uint8)
        f = lib().mz_shard_set_result
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        _check(f(self.comm.ptr, self.ptr, result.ctypes.data, result.size), "mz_shard_set_result")

    def align(self):
        """the product's compute on the share where it lies in HBM (mz_link_plan + mz_link_finish)"""
        f = lib().mz_shard_align
        f.argtypes = [C.c_void_p]
        _check(f(self.ptr), "mz_shard_align")

    def totals(self):
        """(band cells, pairs without a result) of this rank's share, once it has a result image"""
        a, b = C.c_int64(0), C.c_int64(0)
        f = lib().mz_shard_totals
        f.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        _check(f(self.comm.ptr, self.ptr, C.byref(a), C.byref(b)), "mz_shard_totals")
        return a.value, b.value

    def gather(self):
        """every rank calls; the root gets (outs: OUT_DT records of the whole list in the jobs' order, pairs without a result)"""
        f = lib().mz_shard_gather
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        is_root = self.comm.rank == self.root
        outs = np.zeros(len(self.jobs), dtype=OUT_DT) if is_root else None
        rc = f(self.comm.ptr, self.root, self.ptr, self.jobs.ctypes.data if is_root and len(self.jobs) else None,
               outs.ctypes.data if is_root and len(outs) else None)
        try:
            _check(rc, "mz_shard_gather")
        except RuntimeError:
            if is_root and len(outs):                      # (shares assembled before the one that failed own their blocks: include/mz_shard.h)
                free_outs(outs)
            raise
        return (outs, rc) if is_root else (None, 0)

    def free(self):
        if self.ptr:
            f = lib().mz_shard_free
            f.argtypes = [C.c_void_p, C.c_void_p]
            f(self.comm.ptr, self.ptr)
            self.ptr = None


class ShardTimes(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("pack_s", "comm_s", "align_s", "assemble_s", "wall_s")] + \
               [("chunks", C.c_int), ("steps", C.c_int)] + [(k, C.c_int64) for k in ("pairs", "cells", "failed")]


_ALIGN_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(LinkDesc), C.c_void_p, C.c_void_p, C.c_void_p)


def shard_run(comm: Comm, root: int, jobs: Optional[np.ndarray] = None, chunks: int = 0, align=None):
    """mz_shard_run(): the whole exchange of a list that exists on `root`, in chunks that overlap -- packed, moved, aligned and assembled
    side by side (include/mz_shard.h).  Every rank calls; `jobs`: the root's list (JOB_DT records), None elsewhere.  `align`: None = the
    library's GPU path; else a function (chunk, desc: int64[8], image: uint8[], exc: uint8[]) -> result image (uint8[]) that stands in
    for it (the CPU tests: the oracle).  Returns (outs, failed, times): outs = OUT_DT records of the whole list in the jobs' order on
    the root (release with free_outs()), None elsewhere; times = dict of this rank's mz_shard_times."""
    L = lib()
    is_root = comm.rank == root
    jobs = np.ascontiguousarray(jobs) if is_root and jobs is not None else None
    n = len(jobs) if jobs is not None else 0
    outs = np.zeros(n, dtype=OUT_DT) if is_root else None
    err = []

    def c_align(_user, chunk, dptr, image, exc, handle):
        try:
            d = dptr.contents
            desc = np.array([getattr(d, k) for k, _ in LinkDesc._fields_], dtype=np.int64)
            view = lambda ptr, nb: np.ctypeslib.as_array((C.c_uint8 * nb).from_address(ptr)).copy() if nb else np.zeros(0, np.uint8)  # noqa: E731
            res = np.ascontiguousarray(align(int(chunk), desc, view(image, int(d.image_bytes)), view(exc, int(d.exc_bytes))), dtype=np.uint8)
            f = L.mz_shard_chunk_result
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
            return 0 if f(handle, res.ctypes.data, res.size) == 0 else -1
        except Exception as e:                              # noqa: BLE001  (into the library's error return; re-raised below)
            err.append(e)
            return -1
    cb = _ALIGN_T(c_align) if align is not None else C.cast(None, _ALIGN_T)
    tm = ShardTimes()
    f = L.mz_shard_run
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, _ALIGN_T, C.c_void_p, C.POINTER(ShardTimes)]
    rc = f(comm.ptr, root, n, jobs.ctypes.data if n else None, outs.ctypes.data if n else None, int(chunks), cb, None, C.byref(tm))
    times = {k: getattr(tm, k) for k, _ in ShardTimes._fields_}
    if rc < 0:
        if is_root and n:                                    # (chunks assembled before the failure own their blocks)
            free_outs(outs)
        if err:
            raise err[0]
        _check(rc, "mz_shard_run")
    return outs, (rc if is_root else 0), times


def shard_traffic():
    a, b = C.c_int64(0), C.c_int64(0)
    lib().mz_shard_traffic(C.byref(a), C.byref(b))
    return a.value, b.value


def link_expand(desc: np.ndarray, image, exc) -> dict:
    """mz_link_expand(): the image as a device-resident batch's tensors (DevBatch.from_tensors takes them): the header arrays are
    views into the image, the pools are expanded beside it (byte classes as canonical letters, int32 bounds)."""
    import torch
    torch.cuda.set_device(image.device)
    d, at = _desc(desc), link_parts(desc)
    n, dev = int(d.n), image.device
    na, nb = _al256(d.colsA // 2), _al256(d.colsB // 2)
    cols = torch.empty(2 * (na + nb) + 256, dtype=torch.uint8, device=dev)
    LB = torch.empty(max(int(d.band), 1), dtype=torch.int32, device=dev)
    RB = torch.empty(max(int(d.band), 1), dtype=torch.int32, device=dev)
    f = lib().mz_link_expand
    f.argtypes = [C.POINTER(LinkDesc)] + [C.c_void_p] * 6
    _check(f(C.byref(d), image.data_ptr(), exc.data_ptr() if exc.numel() else None, cols.data_ptr(), LB.data_ptr(), RB.data_ptr(),
             torch.cuda.current_stream(dev).cuda_stream), "mz_link_expand")
    v = lambda k, dt, sz: image[at[k]: at[k] + sz * n].view(dt)  # noqa: E731
    t = {k: v(k, torch.int32, 4) for k in ("K", "L", "M", "N")}
    t.update({k: v(k, torch.int64, 8) for k in ("offA", "offB", "offBand")})
    t.update(poolA=cols[: 2 * na], poolB=cols[2 * na:], poolLB=LB, poolRB=RB)
    return t


class PreJob(C.Structure):
    _fields_ = [("K", C.c_int), ("L1", C.c_int), ("M_all", C.c_int), ("N_all", C.c_int), ("radius", C.c_int),
                ("rows1", C.POINTER(C.c_char_p)), ("rows2", C.POINTER(C.c_char_p)), ("v", C.c_int)]


class PreOut(C.Structure):
    _fields_ = [("status", C.c_int), ("badrow", C.c_int), ("null_result", C.c_int), ("stage", C.c_int), ("M", C.c_int), ("N", C.c_int),
                ("OM", C.c_int), ("score", C.c_double), ("size", C.POINTER(C.c_int)), ("rows", C.c_void_p), ("block", C.c_void_p)]


def preyama_batch(jobs: Sequence[tuple]):
    """jobs: sequence of (rows1, rows2, radius[, v]) -- rows1: the K rows of the first block over the overlap (bytes, equal
    lengths), rows2: the rows of the second block over the overlap, its top (reference) row first; v = 1 (default): one-stage
    merge, v = 0: two stages.  Through mz_preyama_batch(): text in, text out, everything between on the GPU.  Returns a list of dicts
    (status, null, M, N, OM, score, size, rows) -- rows as a list of bytes."""
    n = len(jobs)
    cj, co, keep = (PreJob * n)(), (PreOut * n)(), []
    for i, job in enumerate(jobs):
        r1, r2, radius = job[:3]
        a1, a2 = (C.c_char_p * len(r1))(*r1), (C.c_char_p * len(r2))(*r2)
        keep += [a1, a2]
        cj[i].K, cj[i].L1, cj[i].M_all, cj[i].N_all, cj[i].radius = len(r1), len(r2), len(r1[0]), len(r2[0]), radius
        cj[i].v = job[3] if len(job) > 3 else 1
        cj[i].rows1, cj[i].rows2 = a1, a2
    f = lib().mz_preyama_batch
    f.argtypes = [C.c_int, C.POINTER(PreJob), C.POINTER(PreOut)]
    fr = lib().mz_free_preouts
    fr.argtypes = [C.c_int, C.POINTER(PreOut)]
    rc = f(n, cj, co)
    try:
        _check(rc, "mz_preyama_batch")                   # (an error return leaves the blocks of the chunks that did come back: freed below)
        out = []
        for i in range(n):
            o, W = co[i], cj[i].K + cj[i].L1 - 1
            d = dict(status=o.status, badrow=o.badrow, null=bool(o.null_result), null_code=o.null_result, stage=o.stage, M=o.M, N=o.N, OM=o.OM,
                     score=o.score, size=None, rows=None)
            if o.rows:
                d["size"] = [o.size[k] for k in range(W)]
                raw = C.string_at(o.rows, W * o.OM)
                d["rows"] = [raw[k * o.OM:(k + 1) * o.OM] for k in range(W)]
            out.append(d)
    finally:
        fr(n, co)                                        # the call's result blocks (one per chunk: mz_preout.block)
    return out


def preyama_batch_records(jobs: np.ndarray, outs: np.ndarray) -> int:
    """mz_preyama_batch() on record arrays (synth.make_pre_batch); the caller reads outs and then calls free_preouts()"""
    f = lib().mz_preyama_batch
    f.argtypes = [C.c_int, C.POINTER(PreJob), C.POINTER(PreOut)]
    rc = f(len(jobs), jobs.ctypes.data_as(C.POINTER(PreJob)), outs.ctypes.data_as(C.POINTER(PreOut)))
    _check(rc, "mz_preyama_batch")
    return rc


def free_preouts(outs: np.ndarray):
    f = lib().mz_free_preouts
    f.argtypes = [C.c_int, C.POINTER(PreOut)]
    f(len(outs), outs.ctypes.data_as(C.POINTER(PreOut)))


def pre_link_bytes():
    """(bytes up, bytes down, band cells) of the last mz_preyama_batch() call"""
    up, down, cells = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    lib().mz_pre_link_bytes(C.byref(up), C.byref(down), C.byref(cells))
    return up.value, down.value, cells.value


def preout_rows(outs: np.ndarray, i: int, W: int):
    """rows (list of bytes) and base counts of merge i of a finished call, None without a block"""
    o = outs[i]
    if not o["rows"]:
        return None, None
    om = int(o["OM"])
    raw = C.string_at(int(o["rows"]), W * om)
    size = np.ctypeslib.as_array(C.cast(int(o["size"]), C.POINTER(C.c_int)), shape=(W,)).copy()
    return [raw[k * om:(k + 1) * om] for k in range(W)], size


def yama_one(A, B, LB, RB) -> Result:
    return yama_batch([(A, B, LB, RB)])[0]


class DevBatch:
    """A device-resident batch built from torch tensors (torch is plumbing: HBM allocations and the
    current HIP stream).  Inputs are packed pools (see include/mz_amd.h)."""

    def __init__(self, host: dict, device="cuda:0", cap_tb: Optional[int] = None, _tensors: Optional[dict] = None):
        import torch
        self.torch = torch
        self.dev = torch.device(device)
        if self.dev.index is not None:
            torch.cuda.set_device(self.dev)              # the library launches on the current device's streams
        if _tensors is not None:                         # already on the device (from_tensors)
            self.t = _tensors
            n = int(self.t["K"].numel())
        else:
            n = len(host["K"])
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.dev)  # noqa: E731
            self.t = {k: t(host[k], np.int32) for k in ("K", "L", "M", "N", "poolLB", "poolRB")}
            self.t.update({k: t(host[k], np.int64) for k in ("offA", "offB", "offBand")})
            self.t.update({k: t(host[k], np.uint8) for k in ("poolA", "poolB")})
        self.n = n
        self.plan_mem = torch.empty(lib().mz_dev_plan_bytes(n), dtype=torch.uint8, device=self.dev)
        self.c = DevBatchC()
        self.c.n = n
        for k, v in self.t.items():
            setattr(self.c, k, v.data_ptr())
        lib().mz_dev_carve(C.byref(self.c), self.plan_mem.data_ptr())
        # Workspace sizes come from the plan itself (what mz_yama_batch() does on the host path): plan once with
        # unlimited capacities, read the totals, allocate.  They hold for the kernel selection in force now
        # (mz_enable_fast / mz_enable_row); cap_tb overrides the traceback size (tests of MZ_E_WORKSPACE).
        self.c.capTb = self.c.capScript = self.c.capOut = self.c.capPrep = 1 << 62
        _check(lib().mz_dev_plan(C.byref(self.c), self.stream_ptr()), "mz_dev_plan")
        torch.cuda.synchronize(self.dev)
        tot = self._view(self.c.totals, 8, np.int64)
        tot16 = np.ascontiguousarray(self._view(self.c.totals, 16, np.int64))
        lib().mz_walk_choice.argtypes = [C.c_int, C.c_void_p]
        self.c.walk_hint = lib().mz_walk_choice(n, tot16.ctypes.data)     # (pipelined form: one walk launch instead of two)
        lib().mz_dp_hint.argtypes = [C.c_int, C.c_void_p]
        self.c.dp_hint = lib().mz_dp_hint(n, tot16.ctypes.data)           # (only the DP kernels that have pairs, side by side)
        lib().mz_dp_grid.argtypes = [C.c_int, C.c_void_p]
        self.c.dp_grid = lib().mz_dp_grid(n, tot16.ctypes.data)           # (and no more waves than pairs for the counter kernels)
        lib().mz_dp_rows.argtypes = [C.c_int, C.c_void_p]
        self.c.dp_rows = lib().mz_dp_rows(n, tot16.ctypes.data)           # (nor more blocks than row-parallel pairs)
        # the hints hold for the kernel selection and scores in force NOW; the library ignores them (every kernel launched,
        # as with dp_hint = 0) if mz_enable_row / mz_enable_fast / mz_set_scores changed things before run()
        self.c.hint_gen = lib().mz_hint_generation()
        tb = int(tot[0]) if cap_tb is None else int(cap_tb)
        sc, ou, pr = int(tot[1]), int(tot[2]), int(tot[4])
        self.tbw = torch.empty(tb + 64, dtype=torch.int32, device=self.dev)
        self.script = torch.empty(sc + 64, dtype=torch.uint8, device=self.dev)
        self.out = torch.empty(ou + 64, dtype=torch.uint8, device=self.dev)
        self.prep = torch.empty(pr + 64, dtype=torch.int32, device=self.dev)   # transposed band bounds of the COL pairs
        self.c.tbw, self.c.script, self.c.out, self.c.prep = self.tbw.data_ptr(), self.script.data_ptr(), self.out.data_ptr(), self.prep.data_ptr()
        self.c.capTb, self.c.capScript, self.c.capOut, self.c.capPrep = tb + 64, sc + 64, ou + 64, pr + 64
        torch.cuda.synchronize(self.dev)                     # inputs complete (run_async's plan runs on a library stream)

    @classmethod
    def from_tensors(cls, tensors: dict, device=None) -> "DevBatch":
        """a batch whose packed pools are ALREADY device tensors (e.g. received over RCCL by multiz_amd.shard):
        K, L, M, N, poolLB, poolRB int32; offA, offB, offBand int64; poolA, poolB uint8 -- used in place"""
        import torch
        want = dict(K=torch.int32, L=torch.int32, M=torch.int32, N=torch.int32, poolLB=torch.int32, poolRB=torch.int32,
                    offA=torch.int64, offB=torch.int64, offBand=torch.int64, poolA=torch.uint8, poolB=torch.uint8)
        dev = torch.device(device) if device is not None else tensors["K"].device
        if dev.type != "cuda":
            raise RuntimeError("DevBatch needs HIP device tensors (there is no CPU path in the product)")
        t = {k: tensors[k].to(device=dev, dtype=dt).contiguous() for k, dt in want.items()}
        for k in ("poolA", "poolB", "poolLB", "poolRB"):       # (an empty pool still needs an address)
            if t[k].numel() == 0:
                t[k] = torch.zeros(1, dtype=want[k], device=dev)
        return cls(None, device=dev, _tensors=t)

    def results_device(self) -> dict:
        """the plan/result arrays as device tensors (views into plan_mem): nothing crosses PCIe"""
        self.torch.cuda.synchronize(self.dev)
        n, torch = self.n, self.torch

        def v(ptr, count, dt):
            off = ptr - self.plan_mem.data_ptr()
            return self.plan_mem[off: off + count * dt.itemsize].view(dt)
        return dict(status=v(self.c.status, n, torch.int32), mode=v(self.c.mode, n, torch.int32), cells=v(self.c.cells, n, torch.int64),
                    om=v(self.c.om, n, torch.int32), offOut=v(self.c.offOut, n, torch.int64), totals=v(self.c.totals, 8, torch.int64))

    def alternate(self) -> "DevBatch":
        """a second workspace (traceback, script, output, plan arrays) over the SAME resident inputs, for the
        pipelined mz_dev_run_async(): consecutive batches must not share a workspace"""
        import copy
        torch = self.torch
        o = copy.copy(self)
        o.plan_mem = torch.empty_like(self.plan_mem)
        o.c = DevBatchC()
        C.memmove(C.byref(o.c), C.byref(self.c), C.sizeof(DevBatchC))
        lib().mz_dev_carve(C.byref(o.c), o.plan_mem.data_ptr())
        o.tbw, o.script, o.out, o.prep = (torch.empty_like(t) for t in (self.tbw, self.script, self.out, self.prep))
        o.c.tbw, o.c.script, o.c.out, o.c.prep = o.tbw.data_ptr(), o.script.data_ptr(), o.out.data_ptr(), o.prep.data_ptr()
        return o

    def run_async(self, ready_event=None):
        """inputs must be complete (e.g. torch.cuda.synchronize() after building the batch) or ready_event given"""
        ev = None if ready_event is None else ready_event.cuda_event
        _check(lib().mz_dev_run_async(C.byref(self.c), self.stream_ptr(), ev), "mz_dev_run_async")

    def wait(self):
        _check(lib().mz_dev_wait(self.stream_ptr()), "mz_dev_wait")

    def stream_ptr(self):
        return self.torch.cuda.current_stream(self.dev).cuda_stream

    def run(self, timed: bool = False):
        ms = (C.c_float * 4)() if timed else None
        _check(lib().mz_dev_run(C.byref(self.c), self.stream_ptr(), ms), "mz_dev_run")
        return list(ms) if timed else None

    def _view(self, ptr, n, dtype):
        # plan arrays live inside plan_mem; build views by offset
        off = ptr - self.plan_mem.data_ptr()
        nbytes = n * np.dtype(dtype).itemsize
        return self.plan_mem[off: off + nbytes].cpu().numpy().view(dtype)

    def results(self):
        self.torch.cuda.synchronize(self.dev)
        n = self.n
        return dict(status=self._view(self.c.status, n, np.int32), mode=self._view(self.c.mode, n, np.int32),
                    cells=self._view(self.c.cells, n, np.int64), om=self._view(self.c.om, n, np.int32),
                    final3=self._view(self.c.final3, 3 * n, np.int32).reshape(n, 3),
                    offOut=self._view(self.c.offOut, n, np.int64), totals=self._view(self.c.totals, 8, np.int64),
                    offPrep=self._view(self.c.offPrep, n, np.int64), edgeLo=self._view(self.c.edgeLo, n, np.int32),
                    edgeHi=self._view(self.c.edgeHi, n, np.int32))

    def output_cols(self, i: int, res: dict, K: int, L: int):
        om, off = int(res["om"][i]), int(res["offOut"][i])
        return self.out[off: off + om * (K + L)].cpu().numpy().reshape(om, K + L)
