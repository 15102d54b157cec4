"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercising partition, grouped
send/recv scatter, gather + reassembly and the closing all-reduce of multiz_amd/shard.py.  The per-rank
compute is the CPU oracle standing in for the device (this test checks the exchange, not the kernel; the same
exchange with the real device compute runs under -m gpu in test_shard_gpu.py)."""
import os
import socket

import numpy as np
import pytest

import inputs
from oracle import mzoracle as mo


def _batch(seed, n):
    rng = np.random.default_rng(seed)
    pairs = []
    while len(pairs) < n:
        K, L = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        M, N = int(rng.integers(20, 160)), int(rng.integers(20, 160))
        A, B, LB, RB = inputs.make_pair(rng, K, L, M, N, 30, "diag", mo.smooth)
        if mo.check(M, N, LB, RB)[0] == 0:
            pairs.append((A, B, LB, RB))
    K = np.array([p[0].shape[1] for p in pairs], np.int32); L = np.array([p[1].shape[1] for p in pairs], np.int32)
    M = np.array([p[0].shape[0] for p in pairs], np.int32); N = np.array([p[1].shape[0] for p in pairs], np.int32)
    cs = lambda xs: np.concatenate([[0], np.cumsum(xs)[:-1]]).astype(np.int64)  # noqa: E731
    return dict(K=K, L=L, M=M, N=N, offA=cs([p[0].size for p in pairs]), offB=cs([p[1].size for p in pairs]),
                offBand=cs([p[2].size for p in pairs]), poolA=np.concatenate([p[0].ravel() for p in pairs]),
                poolB=np.concatenate([p[1].ravel() for p in pairs]),
                poolLB=np.concatenate([p[2] for p in pairs]).astype(np.int32), poolRB=np.concatenate([p[3] for p in pairs]).astype(np.int32)), pairs


def _oracle_compute(shard):
    """stands in for multiz_amd.shard.device_compute on a box without a GPU: same contract (om, status, off, out as
    torch tensors; out padded between pairs like the device's output buffer)"""
    import torch
    shard = {k: v.cpu().numpy() for k, v in shard.items()}
    n = len(shard["K"])
    om, off, chunks, cells, pos = np.zeros(n, np.int32), np.zeros(n, np.int64), [], 0, 0
    for i in range(n):
        K, L, M, N = (int(shard[k][i]) for k in ("K", "L", "M", "N"))
        a0, b0, d0 = int(shard["offA"][i]), int(shard["offB"][i]), int(shard["offBand"][i])
        r = mo.yama(shard["poolA"][a0:a0 + K * M].reshape(M, K), shard["poolB"][b0:b0 + L * N].reshape(N, L),
                    shard["poolLB"][d0:d0 + M + 1], shard["poolRB"][d0:d0 + M + 1], variant="profile")
        om[i], off[i] = r.OM, pos
        pad = (-r.cols.size) % 16 + 16                       # the device pads its slices; the gather must not care
        chunks += [r.cols.ravel(), np.full(pad, 0xEE, np.uint8)]
        pos += r.cols.size + pad
        cells += mo.band_cells(shard["poolLB"][d0:d0 + M + 1], shard["poolRB"][d0:d0 + M + 1])
    out = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    return dict(om=torch.from_numpy(om), status=torch.zeros(n, dtype=torch.int32), off=torch.from_numpy(off),
                out=torch.from_numpy(out), cells=cells, failed=0)


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    from multiz_amd import shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch, pairs = _batch(11, n) if rank == 0 else (None, None)
        sh, totals = shard.run_sharded(batch, _oracle_compute)
        if rank == 0:
            ok = bool((sh.status == 0).all()) and sorted(set(sh.owner.tolist())) == list(range(min(world, n)))
            for i, (A, B, LB, RB) in enumerate(pairs):
                want = mo.yama(A, B, LB, RB)
                ok &= int(sh.om[i]) == want.OM and np.array_equal(sh.cols(i), want.cols.ravel())
            cells = sum(mo.band_cells(p[2], p[3]) for p in pairs)
            q.put((ok, totals, (n, cells, 0)))
    finally:
        dist.destroy_process_group()


def _link_worker(rank, world, port, n, q):
    """the same exchange on link images (shard.run_sharded_link): class nibbles + band steps out, records + 2-bit scripts back, the
    root assembles from its own pools; tests/linkfmt.py decodes / encodes where the product runs its kernels"""
    import torch.distributed as dist
    import linkfmt
    from multiz_amd import shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch, pairs = _batch(11, n) if rank == 0 else (None, None)
        sh, totals = shard.run_sharded_link(batch, linkfmt.oracle_compute)
        if rank == 0:
            ok = bool((sh.status == 0).all()) and sorted(set(sh.owner.tolist())) == list(range(min(world, n)))
            for i, (A, B, LB, RB) in enumerate(pairs):
                want = mo.yama(A, B, LB, RB)
                ok &= int(sh.om[i]) == want.OM and np.array_equal(sh.cols(i), want.cols.ravel())
            cells = sum(mo.band_cells(p[2], p[3]) for p in pairs)
            pools = sum(batch[k].nbytes for k in ("poolA", "poolB", "poolLB", "poolRB"))
            merged = int(sum(int(sh.om[i]) * (p[0].shape[1] + p[1].shape[1]) for i, p in enumerate(pairs)))
            ex = dict(shard.last_exchange)
            ok &= ex["pairs"] == n and ex["down_bytes"] < merged + 200 * n + 1024 * world
            ok &= n < 20 or ex["up_bytes"] < pools             # (every part of an image is padded to 256 bytes: a handful of tiny pairs is all padding)
            sh.release()
            q.put((ok, totals, (n, cells, 0)))
    finally:
        dist.destroy_process_group()


def _chunk_worker(rank, world, port, n, q):
    """the exchange in chunks (shard.run_sharded_chunks -> mz_shard_run): a world of processes over gloo, three chunks a rank -- with
    n < ranks x chunks most of them empty --, the oracle behind the align hook"""
    import torch.distributed as dist
    import linkfmt
    from multiz_amd import shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch, pairs = _batch(11, n) if rank == 0 else (None, None)
        sh, totals, times = shard.run_sharded_chunks(batch, lambda ch, desc, image, exc: linkfmt.oracle_result_image(desc, image, exc)[0], chunks=3)
        ok = times["chunks"] == 3 and times["steps"] == 7
        if rank == 0:
            ok &= bool((sh.status == 0).all())
            for i, (A, B, LB, RB) in enumerate(pairs):
                want = mo.yama(A, B, LB, RB)
                ok &= int(sh.om[i]) == want.OM and np.array_equal(sh.cols(i), want.cols.ravel())
            cells = sum(mo.band_cells(p[2], p[3]) for p in pairs)
            merged = int(sum(int(sh.om[i]) * (p[0].shape[1] + p[1].shape[1]) for i, p in enumerate(pairs)))
            ex = dict(shard.last_exchange)
            ok &= ex["pairs"] == n and ex["down_bytes"] < merged + 200 * n + 1024 * world * 3
            sh.release()
            q.put((ok, totals, (n, cells, 0)))
        else:
            assert ok
        shard.drop_comms()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world,n,worker", [(2, 23, "pools"), (3, 10, "pools"), (2, 1, "pools"), (2, 23, "link"), (3, 10, "link"), (2, 1, "link"),
                                            (2, 23, "chunks"), (3, 10, "chunks"), (2, 1, "chunks")])
def test_scatter_compute_gather(world, n, worker):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target={"pools": _worker, "link": _link_worker, "chunks": _chunk_worker}[worker], args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _q
    for _ in range(180):
        try:
            ok, totals, want = q.get(timeout=1)
            break
        except _q.Empty:
            assert all(p.exitcode in (None, 0) for p in procs), "a rank died"
    else:
        raise AssertionError("timed out")
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok, "re-assembled results differ from the unsharded oracle"
    assert totals == want          # pairs, cells, failures summed over ranks by the closing all-reduce


def test_partition_is_balanced_and_complete():
    from multiz_amd import shard
    rng = np.random.default_rng(0)
    cost = rng.integers(1, 10**6, size=1000)
    for world in (1, 2, 8):
        parts = shard.partition(cost, world)
        assert sorted(np.concatenate(parts).tolist()) == list(range(1000))
        assert all((np.diff(p) > 0).all() for p in parts)       # ascending within a rank
        loads = np.array([cost[p].sum() for p in parts])
        assert loads.max() - loads.min() <= cost.max()          # the snake deal's bound (as for LPT)
    # a million pairs are dealt out without a per-pair loop: well under a second
    import time
    big = rng.integers(10**4, 10**5, size=1_000_000)
    t = time.perf_counter()
    parts = shard.partition(big, 8)
    assert time.perf_counter() - t < 2.0
    loads = np.array([big[p].sum() for p in parts])
    assert (loads.max() - loads.min()) / loads.mean() < 1e-4
    batch, pairs = _batch(3, 12)
    sub = shard.take(batch, np.array([7, 2, 9]))
    assert np.array_equal(shard.pair_cost(sub), shard.pair_cost(batch)[[7, 2, 9]])
    from multiz_amd import synth
    for j, i in enumerate((7, 2, 9)):                            # the re-packed pairs are the originals, byte for byte
        for x, y in zip(synth.pair_of(sub, j), pairs[i]):
            assert np.array_equal(x, y)
    assert len(shard.take(batch, np.zeros(0, np.int64))["poolA"]) == 0
