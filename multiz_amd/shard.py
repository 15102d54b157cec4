"""Sharding a batch of independent block pairs over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).
Block pairs share nothing, so the data path has no collective: the root partitions the work list by
cost (band cells, heaviest first, dealt out in a snake so that every rank gets the same mix), hands every
rank its packed sub-batch with one group of point-to-point sends (xGMI is point-to-point: one peer per
link, all links busy at once), each rank aligns its shard ON ITS GPU straight from the received tensors
(`device_compute`: a DevBatch over them, no host bounce), results come back the same way and are addressed
in the caller's order.  A single all-reduce of three scalars (pairs, cells, failures) closes the batch.

Nothing here loops over pairs in Python: partition is a sort, re-packing is a C gather of segments
(mz_gather_segments in libmzamd.so), re-assembly is index arithmetic -- a guide-tree level of a million
merges (BASELINE config 4) is dealt out in a fraction of the time its alignment takes.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional

import numpy as np

_I32 = ("K", "L", "M", "N", "poolLB", "poolRB")
_I64 = ("offA", "offB", "offBand")
_U8 = ("poolA", "poolB")
FIELDS = _I32 + _I64 + _U8


def pair_cost(batch: Dict[str, np.ndarray]) -> np.ndarray:
    """band cells per pair = the reference's tback_size (mz_yama.c:60-66); the DP cost is linear in it"""
    M = batch["M"].astype(np.int64)
    off = batch["offBand"].astype(np.int64)
    w = batch["poolRB"].astype(np.int64) - batch["poolLB"].astype(np.int64) + 1
    cs = np.concatenate([[0], np.cumsum(w)])
    return cs[off + M + 1] - cs[off]


def partition(cost: np.ndarray, world: int) -> List[np.ndarray]:
    """Heaviest pair first, dealt to the ranks in a snake (0..W-1, W-1..0, ...): one sort, no per-pair loop.
    Within a round of 2W pairs rank i gets the i-th heaviest and the i-th lightest, so the loads differ by at
    most the spread of the costs (max - min), like the longest-processing-time greedy this replaces.
    Indices of each rank come back in ascending order."""
    n = len(cost)
    order = np.argsort(-np.asarray(cost, dtype=np.int64), kind="stable")
    pos = np.arange(n, dtype=np.int64)
    rnd, k = pos // world, pos % world
    owner_sorted = np.where(rnd % 2 == 0, k, world - 1 - k)
    owner = np.empty(n, dtype=np.int64)
    owner[order] = owner_sorted
    return [np.flatnonzero(owner == r) for r in range(world)]


def _gather(pool: np.ndarray, off: np.ndarray, ln: np.ndarray) -> np.ndarray:
    """pool[off[i] : off[i]+ln[i]] back to back, copied by the library (memcpy per segment on the host threads)"""
    from .api import lib
    off = np.ascontiguousarray(off, dtype=np.int64)
    ln = np.ascontiguousarray(ln, dtype=np.int64)
    out = np.empty(int(ln.sum()), dtype=pool.dtype)
    if len(off):
        pos = np.ascontiguousarray(np.cumsum(ln) - ln, dtype=np.int64)
        src = np.ascontiguousarray(pool)
        f = lib().mz_gather_segments
        f.restype = C.c_int
        f.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if f(len(off), pool.dtype.itemsize, off.ctypes.data, ln.ctypes.data, pos.ctypes.data, src.ctypes.data, out.ctypes.data) != 0:
            raise RuntimeError("mz_gather_segments failed (bad arguments)")
    return out


def _excl(x: np.ndarray) -> np.ndarray:
    return (np.cumsum(x) - x).astype(np.int64)


def take(batch: Dict[str, np.ndarray], idx: np.ndarray) -> Dict[str, np.ndarray]:
    """re-pack the chosen pairs as a batch of their own"""
    idx = np.asarray(idx, dtype=np.int64)
    K, L, M, N = (batch[k][idx].astype(np.int64) for k in ("K", "L", "M", "N"))
    la, lb, ld = K * M, L * N, M + 1
    out = {k: batch[k][idx].astype(np.int32) for k in ("K", "L", "M", "N")}
    out.update(offA=_excl(la), offB=_excl(lb), offBand=_excl(ld),
               poolA=_gather(batch["poolA"], batch["offA"][idx], la), poolB=_gather(batch["poolB"], batch["offB"][idx], lb),
               poolLB=_gather(batch["poolLB"], batch["offBand"][idx], ld), poolRB=_gather(batch["poolRB"], batch["offBand"][idx], ld))
    return out


def _dtype(name):
    return np.int32 if name in _I32 else np.int64 if name in _I64 else np.uint8


def _tdtype(torch, name):
    return getattr(torch, np.dtype(_dtype(name)).name)


def scatter_batch(batch: Optional[Dict[str, np.ndarray]], src: int = 0, device="cpu", group=None):
    """Root: partition + send each rank its shard.  Every rank: returns (shard as torch tensors on `device`, global
    indices of its pairs as a numpy array).  Sizes travel in one broadcast header; payloads in ONE batch of
    point-to-point ops (RCCL: one grouped launch, all xGMI links at once)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    header = torch.zeros((world, len(FIELDS) + 1), dtype=torch.int64, device=device)
    shards, parts = None, None
    if rank == src:
        parts = partition(pair_cost(batch), world)
        shards = [take(batch, p) for p in parts]
        header[:, :-1] = torch.tensor([[len(s[f]) for f in FIELDS] for s in shards], dtype=torch.int64)
        header[:, -1] = torch.tensor([len(p) for p in parts], dtype=torch.int64)
    dist.broadcast(header, src=src, group=group)
    sizes = header[rank].tolist()
    ops, keep = [], []
    if rank == src:
        mine, my_idx = None, None
        for r in range(world):
            tens = {f: torch.from_numpy(np.ascontiguousarray(shards[r][f], dtype=_dtype(f))).to(device) for f in FIELDS}
            tidx = torch.from_numpy(parts[r].astype(np.int64)).to(device)
            if r == src:
                mine, my_idx = tens, tidx
            else:
                keep.append((tens, tidx))
                ops += [dist.P2POp(dist.isend, tens[f], r, group) for f in FIELDS if tens[f].numel()]
                if tidx.numel():
                    ops.append(dist.P2POp(dist.isend, tidx, r, group))
    else:
        mine = {f: torch.empty(sizes[i], dtype=_tdtype(torch, f), device=device) for i, f in enumerate(FIELDS)}
        my_idx = torch.empty(sizes[-1], dtype=torch.int64, device=device)
        ops += [dist.P2POp(dist.irecv, mine[f], src, group) for f in FIELDS if mine[f].numel()]
        if my_idx.numel():
            ops.append(dist.P2POp(dist.irecv, my_idx, src, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return mine, my_idx.cpu().numpy()


class Sharded:
    """What the root holds after the gather: every rank's output buffer as it left the GPU, and for each pair of the
    caller's list its owner, its offset in that buffer and its merged length -- the caller's order by indexing."""

    def __init__(self, n_total: int, widths: np.ndarray):
        self.om = np.zeros(n_total, dtype=np.int32)
        self.status = np.full(n_total, -1, dtype=np.int32)
        self.owner = np.full(n_total, -1, dtype=np.int32)
        self.off = np.zeros(n_total, dtype=np.int64)
        self.widths = np.asarray(widths, dtype=np.int64)
        self.bufs: Dict[int, np.ndarray] = {}

    def cols(self, i: int) -> np.ndarray:
        o, nb = int(self.off[i]), int(self.om[i]) * int(self.widths[i])
        return self.bufs[int(self.owner[i])][o: o + nb]

    def __iter__(self):
        return (self.cols(i) for i in range(len(self.om)))


def gather_results(res: dict, my_idx: np.ndarray, n_total: int, widths: Optional[np.ndarray], dst: int = 0, device="cpu", group=None):
    """res: what compute returned -- om int32[n], status int32[n], off int64[n] (offset of pair i's merged columns in
    out), out uint8[...] as torch tensors (on `device`, or anywhere: they are moved).  One grouped exchange; the root
    returns a Sharded, the others None."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    t_om, t_st, t_off, t_out = (res[k].to(device) for k in ("om", "status", "off", "out"))
    t_idx = torch.from_numpy(np.ascontiguousarray(my_idx, dtype=np.int64)).to(device)
    meta = torch.tensor([t_om.numel(), t_out.numel()], dtype=torch.int64, device=device)
    metas = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    if rank != dst:
        ops = [dist.P2POp(dist.isend, t, dst, group) for t in (t_om, t_st, t_off, t_idx, t_out) if t.numel()]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return None
    bufs, ops = {}, []
    for r in range(world):
        if r == dst:
            bufs[r] = (t_om, t_st, t_off, t_idx, t_out)
            continue
        n, nb = (int(x) for x in metas[r].tolist())
        b = (torch.empty(n, dtype=torch.int32, device=device), torch.empty(n, dtype=torch.int32, device=device),
             torch.empty(n, dtype=torch.int64, device=device), torch.empty(n, dtype=torch.int64, device=device),
             torch.empty(nb, dtype=torch.uint8, device=device))
        bufs[r] = b
        ops += [dist.P2POp(dist.irecv, t, r, group) for t in b if t.numel()]
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    sh = Sharded(n_total, widths)
    # every rank's buffers to the host: pinned destinations, all copies issued before the one synchronisation (the DMA
    # engine streams them back to back at link rate; a .cpu() per tensor went through pageable staging, one at a time)
    on_gpu = any(t.is_cuda for b in bufs.values() for t in b)
    host = {}
    for r, b in bufs.items():
        if on_gpu:
            dst = tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in b)
            for d_, t in zip(dst, b):
                d_.copy_(t, non_blocking=True)
            host[r] = dst
        else:
            host[r] = b
    if on_gpu:
        torch.cuda.synchronize()
    for r, (om, st, off, idx, out) in host.items():
        i = idx.numpy()
        sh.om[i] = om.numpy()
        sh.status[i] = st.numpy()
        sh.off[i] = off.numpy()
        sh.owner[i] = r
        sh.bufs[r] = out.numpy()
    sh._keep = host                                          # (the numpy views live in these tensors)
    return sh


def close_batch(pairs: int, cells: int, failed: int, device="cpu", group=None):
    """the batch-closing reduction: totals over all ranks"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([pairs, cells, failed], dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    return tuple(int(x) for x in t.tolist())


def device_compute(shard: dict, device=None, keep: Optional[list] = None) -> dict:
    """The product's compute for run_sharded(): the received tensors ARE the device batch (moved only if they arrived
    on another device, as in the gloo tests); plan, DP, walk, emit through the C ABI (mz_dev_run)."""
    import torch
    from .api import DevBatch
    dev = torch.device(device) if device is not None else shard["K"].device
    n = int(shard["K"].numel())
    if n == 0:
        z = lambda dt: torch.zeros(0, dtype=dt, device=dev)  # noqa: E731
        return dict(om=z(torch.int32), status=z(torch.int32), off=z(torch.int64), out=z(torch.uint8), cells=0, failed=0)
    db = DevBatch.from_tensors({k: v.to(dev) for k, v in shard.items()}, device=dev)
    db.run()
    r = db.results_device()
    if keep is not None:
        keep.append(db)
    total = int(r["totals"][2].item())
    return dict(om=r["om"], status=r["status"], off=r["offOut"], out=db.out[:total],
                cells=int(r["cells"].sum().item()), failed=int((r["status"] != 0).sum().item()))


# ------------------------------------------------------------------------------------------------------------------
# The same exchange on LINK IMAGES (include/mz_amd.h, mz_link_*), and IN C (include/mz_shard.h, multiz_amd/csrc/mz_shard.c): what
# leaves the root is what mz_yama_batch() sends over PCIe -- byte classes two per byte and band steps, packed by the library
# straight from the root's pools --, what comes back is a record per pair and the edit scripts at two bits per merged column; the
# root assembles the merged columns from its own A and B.  C2: 3.1 KB out / 0.53 KB back per pair (pools: 6.0 / 4.3 and more), C4's
# tree mix about a third of the pools' 22.8 + 19.5 KB.  Dealing, packing, the grouped sends and receives and the assembly are the
# library's (mz_shard_scatter / _align / _gather); this module only gives it a transport: RCCL where the process group is "nccl"
# (ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd inside the library, the communicator made from an id the group broadcasts),
# the group's own blocking send / recv of host bytes where it is "gloo" (the CPU tests).

last_exchange: Dict[str, int] = {}            # bytes of the last link scatter / gather on this rank (the root's are the totals)
_comms: Dict[int, tuple] = {}                  # id(group object) -> (weak reference to that object, its api.Comm)


def drop_comms():
    """free the transports made so far (before dist.destroy_process_group(): an RCCL communicator must not outlive its job)"""
    for _ref, comm in _comms.values():
        comm.free()
    _comms.clear()


def comm_for(group=None, device="cpu"):
    """the library's transport (api.Comm) for a torch.distributed process group: made once per group OBJECT -- the key is the
    group's id() together with a weak reference to it, so a new group that happens to get a collected one's id (or the default group
    after destroy_process_group() + init_process_group()) makes a new transport and the stale one is freed"""
    import weakref
    import torch
    import torch.distributed as dist
    from . import api
    obj = group if group is not None else dist.distributed_c10d._get_default_group()
    key = id(obj)
    hit = _comms.get(key)
    if hit is not None:
        if hit[0]() is obj:
            return hit[1]
        hit[1].free()                                        # (the id belonged to a group that is gone)
        del _comms[key]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if dist.get_backend(group) == "nccl":
        t = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            t = torch.frombuffer(bytearray(api.Comm.rccl_unique_id()), dtype=torch.uint8).clone()
        t = t.to(device)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        comm = api.Comm.rccl(bytes(t.cpu().numpy().tobytes()), rank, world)
    else:
        def send(buf, peer):
            dist.send(torch.from_numpy(buf), dst=dist.get_global_rank(group, peer) if group is not None else peer, group=group)

        def recv(buf, peer):
            dist.recv(torch.from_numpy(buf), src=dist.get_global_rank(group, peer) if group is not None else peer, group=group)
        comm = api.Comm.custom(rank, world, send, recv)
    try:
        ref = weakref.ref(obj)
    except TypeError:                                        # (a group type without weak references: keep it alive instead)
        ref = (lambda o: (lambda: o))(obj)
    _comms[key] = (ref, comm)
    return comm


def scatter_link(batch: Optional[Dict[str, np.ndarray]], src: int = 0, device="cpu", group=None):
    """Root: deal, pack every rank's jobs as a link image, send (mz_shard_scatter).  Every rank: (its api.Shard, the global indices
    of its pairs)."""
    import torch.distributed as dist
    from . import api
    rank = dist.get_rank(group)
    comm = comm_for(group, device)
    s0, r0 = api.shard_traffic()
    jobs = api.host_jobs(batch)[0] if rank == src else None
    sh = api.Shard(comm, src, jobs)
    s1, r1 = api.shard_traffic()
    if rank == src:
        last_exchange.update(pairs=int(len(batch["K"])), up_bytes=int(s1 - s0))
    return sh, sh.index


def link_compute(sh, device=None) -> dict:
    """The product's compute on a share: where it arrived in HBM over RCCL, mz_shard_align (mz_link_plan / mz_link_finish on the image
    where it lies); where it arrived in host memory (the gloo tests on a GPU box), the image is moved first.  -> dict(cells, failed)"""
    import torch
    from . import api
    try:
        sh.align()
    except RuntimeError as e:
        if "host memory" not in str(e):
            raise
        dev = torch.device(device if device is not None else "cuda")
        image, exc = sh.host_image()
        res = api.link_run(sh.desc, torch.from_numpy(image).to(dev), torch.from_numpy(exc).to(dev))
        torch.cuda.synchronize(dev)                      # (torch's default stream is the NULL stream: the library then runs on its own)
        sh.set_result(res.cpu().numpy())
    cells, failed = sh.totals()
    return dict(cells=cells, failed=failed)


def link_totals(res, n: int) -> dict:
    """cells and failures of a result image from its records, where it lies (torch tensor)"""
    import torch
    from .api import RES_DT
    if n == 0:
        return dict(result=res, cells=0, failed=0)
    rec = res[64: 64 + RES_DT.itemsize * n]
    status = rec.view(torch.int32).view(n, RES_DT.itemsize // 4)[:, 0]
    cells = rec.view(torch.int64).view(n, RES_DT.itemsize // 8)[:, RES_DT.fields["cells"][1] // 8]
    return dict(result=res, cells=int(cells.sum().item()), failed=int((status != 0).sum().item()))


class ShardedOuts:
    """What the root holds after a link gather: outs of the whole list in the caller's order (OUT_DT records: mz_out), the merged
    columns assembled from ITS OWN pools in one block per rank's share.  release() hands the blocks back to the library."""

    def __init__(self, outs: np.ndarray, widths: np.ndarray, index_of_rank: Optional[Dict[int, np.ndarray]] = None):
        self.outs, self.widths = outs, np.asarray(widths, dtype=np.int64)
        self.om, self.status = outs["OM"], outs["status"]
        self.owner = np.full(len(outs), -1, dtype=np.int32)
        for r, idx in (index_of_rank or {}).items():
            self.owner[idx] = r

    def cols(self, i: int) -> np.ndarray:
        nb = int(self.om[i]) * int(self.widths[i])
        return np.ctypeslib.as_array((C.c_uint8 * nb).from_address(int(self.outs["cols"][i]))) if nb else np.zeros(0, np.uint8)

    def __iter__(self):
        return (self.cols(i) for i in range(len(self.om)))

    def release(self):
        from . import api
        if self.outs is not None:
            api.free_outs(self.outs)
            self.outs = None


def gather_link(sh, my_idx: np.ndarray, batch: Optional[Dict[str, np.ndarray]], dst: int = 0, device="cpu", group=None):
    """Every rank's result image to the root (mz_shard_gather), which assembles every share's merged columns from its own pools and
    returns a ShardedOuts; the others None."""
    import torch
    import torch.distributed as dist
    from . import api
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    s0, r0 = api.shard_traffic()
    outs, _failed = sh.gather()
    s1, r1 = api.shard_traffic()
    # (who had which pair: for the callers' checks -- one all-gather of the index lists' sizes and one of the lists)
    n_mine = torch.tensor([len(my_idx)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n_mine, group=group)
    cap = max(int(x.item()) for x in sizes) if world else 0
    mine = torch.full((max(cap, 1),), -1, dtype=torch.int64, device=device)
    mine[: len(my_idx)] = torch.from_numpy(np.ascontiguousarray(my_idx, dtype=np.int64)).to(device)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    if rank != dst:
        last_exchange.update(down_bytes=int(s1 - s0))
        sh.free()
        return None
    last_exchange.update(down_bytes=int(r1 - r0))
    index_of_rank = {r: every[r][: int(sizes[r].item())].cpu().numpy() for r in range(world)}
    res = ShardedOuts(outs, batch["K"].astype(np.int64) + batch["L"].astype(np.int64), index_of_rank)
    sh.free()
    return res


def run_sharded_link(batch: Optional[Dict[str, np.ndarray]], compute: Callable = link_compute, src: int = 0, device="cpu", group=None):
    """scatter_link -> compute(share) -> gather_link.  compute takes the api.Shard, gives it a result image (align() or set_result())
    and returns dict(cells, failed).  Root gets (ShardedOuts, totals); others (None, totals)."""
    sh, my_idx = scatter_link(batch, src, device, group)
    res = compute(sh)
    out = gather_link(sh, my_idx, batch, src, device, group)
    totals = close_batch(len(my_idx), res["cells"], res["failed"], device, group)
    return out, totals


def run_sharded_chunks(batch: Optional[Dict[str, np.ndarray]], align: Optional[Callable] = None, src: int = 0, device="cpu", group=None,
                       chunks: int = 0):
    """The exchange in chunks that overlap (mz_shard_run, include/mz_shard.h): the root packs chunk t+1 and assembles chunk t-3 while
    the transport moves chunk t down and chunk t-2's results up and every rank's GPU aligns chunk t-1.  `align`: None = the library's
    GPU path (mz_link_plan / mz_link_finish; a "gloo" group's host buffers go up and come down inside the step); else a function
    (chunk, desc, image, exc) -> result image (the CPU tests: the oracle).  Root gets (ShardedOuts, totals, times); others (None, totals,
    times); totals = (pairs, cells, failed) over all ranks, times = this rank's mz_shard_times as a dict."""
    import torch.distributed as dist
    from . import api
    rank = dist.get_rank(group)
    comm = comm_for(group, device)
    s0, r0 = api.shard_traffic()
    jobs = api.host_jobs(batch)[0] if rank == src else None
    outs, _failed, times = api.shard_run(comm, src, jobs, chunks=chunks, align=align)
    s1, r1 = api.shard_traffic()
    res = None
    if rank == src:
        last_exchange.update(pairs=int(len(batch["K"])), up_bytes=int(s1 - s0), down_bytes=int(r1 - r0))
        res = ShardedOuts(outs, batch["K"].astype(np.int64) + batch["L"].astype(np.int64))
        res._jobs = jobs                                         # (the merged columns were assembled from these arrays' memory)
    totals = close_batch(int(times["pairs"]), int(times["cells"]), int(times["failed"]), device, group)
    return res, totals, times


def run_sharded(batch: Optional[Dict[str, np.ndarray]], compute: Callable = device_compute, src: int = 0, device="cpu", group=None):
    """scatter -> compute(shard) -> gather.  compute takes the shard (dict of torch tensors on `device`) and returns
    dict(om, status, off, out: torch tensors; cells, failed: ints).  Root gets (Sharded, totals); others (None, totals)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    n_total = torch.tensor([len(batch["K"]) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n_total, src=src, group=group)
    shard, my_idx = scatter_batch(batch, src, device, group)
    res = compute(shard)
    widths = (batch["K"].astype(np.int64) + batch["L"].astype(np.int64)) if rank == src else None
    sh = gather_results(res, my_idx, int(n_total.item()), widths, src, device, group)
    totals = close_batch(len(my_idx), res["cells"], res["failed"], device, group)
    return sh, totals
