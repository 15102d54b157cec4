"""Sharding a batch of independent block pairs over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).
Block pairs share nothing, so the data path has no collective: the root partitions the work list by
cost (longest-processing-time greedy over band cells), hands every rank its packed sub-batch with one
group of point-to-point sends (xGMI is point-to-point: one peer per link, all links busy at once),
each rank aligns its shard, results come back the same way and are put back in the caller's order.
A single all-reduce of three scalars (pairs, cells, failures) closes the batch.
"""
from __future__ import annotations

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
    """LPT greedy: heaviest pair first onto the lightest rank; indices of each rank in ascending order"""
    order = np.argsort(-cost, kind="stable")
    load = np.zeros(world, dtype=np.int64)
    owner = np.empty(len(cost), dtype=np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += int(cost[i])
    return [np.flatnonzero(owner == r) for r in range(world)]


def take(batch: Dict[str, np.ndarray], idx: np.ndarray) -> Dict[str, np.ndarray]:
    """re-pack the chosen pairs as a batch of their own"""
    idx = np.asarray(idx, dtype=np.int64)
    K, L, M, N = (batch[k][idx].astype(np.int64) for k in ("K", "L", "M", "N"))
    la, lb, ld = K * M, L * N, M + 1
    oa, ob, od = (np.concatenate([[0], np.cumsum(x)[:-1]]).astype(np.int64) if len(x) else np.zeros(0, np.int64) for x in (la, lb, ld))

    def gather(pool, off, ln):
        if len(idx) == 0:
            return pool[:0].copy()
        return np.concatenate([pool[int(o): int(o) + int(n)] for o, n in zip(off, ln)])

    out = {k: batch[k][idx].astype(np.int32) for k in ("K", "L", "M", "N")}
    out.update(offA=oa, offB=ob, offBand=od,
               poolA=gather(batch["poolA"], batch["offA"][idx], la), poolB=gather(batch["poolB"], batch["offB"][idx], lb),
               poolLB=gather(batch["poolLB"], batch["offBand"][idx], ld), poolRB=gather(batch["poolRB"], batch["offBand"][idx], ld))
    return out


def _dtype(name):
    return np.int32 if name in _I32 else np.int64 if name in _I64 else np.uint8


def scatter_batch(batch: Optional[Dict[str, np.ndarray]], src: int = 0, device="cpu", group=None):
    """Root: partition + send each rank its shard.  Every rank: returns (shard, global indices of its pairs).
    Sizes travel in one broadcast header; payloads in ONE batch of point-to-point ops."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    header = torch.zeros((world, len(FIELDS) + 1), dtype=torch.int64, device=device)
    shards, parts = None, None
    if rank == src:
        parts = partition(pair_cost(batch), world)
        shards = [take(batch, p) for p in parts]
        for r, s in enumerate(shards):
            header[r, :-1] = torch.tensor([len(s[f]) for f in FIELDS])
            header[r, -1] = len(parts[r])
    dist.broadcast(header, src=src, group=group)
    sizes = header[rank].tolist()
    mine = {f: torch.empty(sizes[i], dtype=getattr(torch, np.dtype(_dtype(f)).name), device=device) for i, f in enumerate(FIELDS)}
    my_idx = torch.empty(sizes[-1], dtype=torch.int64, device=device)
    ops, keep = [], []
    if rank == src:
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
        ops += [dist.P2POp(dist.irecv, mine[f], src, group) for f in FIELDS if mine[f].numel()]
        if my_idx.numel():
            ops.append(dist.P2POp(dist.irecv, my_idx, src, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return {f: mine[f].cpu().numpy() for f in FIELDS}, my_idx.cpu().numpy()


def gather_results(om: np.ndarray, cols: np.ndarray, my_idx: np.ndarray, n_total: int, dst: int = 0, device="cpu", group=None):
    """cols: the rank's merged columns back to back (pair i contributes om[i]*(K+L) bytes).
    Root returns (om_all int32[n_total], list of byte arrays in the caller's pair order); others None."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    meta = torch.tensor([len(om), len(cols)], dtype=torch.int64, device=device)
    metas = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    t_om = torch.from_numpy(np.ascontiguousarray(om, dtype=np.int32)).to(device)
    t_idx = torch.from_numpy(np.ascontiguousarray(my_idx, dtype=np.int64)).to(device)
    t_cols = torch.from_numpy(np.ascontiguousarray(cols, dtype=np.uint8)).to(device)
    if rank != dst:
        ops = [dist.P2POp(dist.isend, t, dst, group) for t in (t_om, t_idx, t_cols) if t.numel()]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return None
    bufs, ops = {}, []
    for r in range(world):
        if r == dst:
            bufs[r] = (t_om, t_idx, t_cols)
            continue
        n, nb = (int(x) for x in metas[r].tolist())
        b = (torch.empty(n, dtype=torch.int32, device=device), torch.empty(n, dtype=torch.int64, device=device),
             torch.empty(nb, dtype=torch.uint8, device=device))
        bufs[r] = b
        ops += [dist.P2POp(dist.irecv, t, r, group) for t in b if t.numel()]
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    om_all = np.zeros(n_total, dtype=np.int32)
    out: List[Optional[np.ndarray]] = [None] * n_total
    return om_all, out, {r: tuple(t.cpu().numpy() for t in b) for r, b in bufs.items()}


def reassemble(gathered, widths: np.ndarray):
    """put every rank's pairs back at their original positions; widths[i] = K+L of pair i"""
    om_all, out, bufs = gathered
    for r, (om, idx, cols) in bufs.items():
        pos = 0
        for m, i in zip(om, idx):
            nb = int(m) * int(widths[i])
            om_all[i] = m
            out[int(i)] = cols[pos: pos + nb]
            pos += nb
    return om_all, out


def close_batch(pairs: int, cells: int, failed: int, device="cpu", group=None):
    """the batch-closing reduction: totals over all ranks"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([pairs, cells, failed], dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    return tuple(int(x) for x in t.tolist())


def run_sharded(batch: Optional[Dict[str, np.ndarray]], compute: Callable, src: int = 0, device="cpu", group=None):
    """scatter -> compute(shard) -> gather.  compute returns (om int32[n], cols uint8 back to back, cells, failed).
    Root gets (om_all, list of merged-column byte arrays in the original order, totals); others (None, None, totals)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    n_total = torch.tensor([len(batch["K"]) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n_total, src=src, group=group)
    shard, my_idx = scatter_batch(batch, src, device, group)
    om, cols, cells, failed = compute(shard)
    g = gather_results(om, cols, my_idx, int(n_total.item()), src, device, group)
    totals = close_batch(len(my_idx), cells, failed, device, group)
    if rank != src:
        return None, None, totals
    widths = batch["K"].astype(np.int64) + batch["L"].astype(np.int64)
    om_all, out = reassemble(g, widths)
    return om_all, out, totals
