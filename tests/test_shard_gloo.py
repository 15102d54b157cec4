"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercising partition, grouped
send/recv scatter, gather + reassembly and the closing all-reduce of multiz_amd/shard.py.  The per-rank
compute is the CPU oracle standing in for the device (this test checks the exchange, not the kernel)."""
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
    n = len(shard["K"])
    om, chunks, cells = np.zeros(n, np.int32), [], 0
    for i in range(n):
        K, L, M, N = (int(shard[k][i]) for k in ("K", "L", "M", "N"))
        a0, b0, d0 = int(shard["offA"][i]), int(shard["offB"][i]), int(shard["offBand"][i])
        r = mo.yama(shard["poolA"][a0:a0 + K * M].reshape(M, K), shard["poolB"][b0:b0 + L * N].reshape(N, L),
                    shard["poolLB"][d0:d0 + M + 1], shard["poolRB"][d0:d0 + M + 1], variant="profile")
        om[i] = r.OM
        chunks.append(r.cols.ravel())
        cells += mo.band_cells(shard["poolLB"][d0:d0 + M + 1], shard["poolRB"][d0:d0 + M + 1])
    return om, (np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)), cells, 0


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    from multiz_amd import shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch, pairs = _batch(11, n) if rank == 0 else (None, None)
        om, out, totals = shard.run_sharded(batch, _oracle_compute)
        if rank == 0:
            ok = True
            for (A, B, LB, RB), m, o in zip(pairs, om, out):
                want = mo.yama(A, B, LB, RB)
                ok &= int(m) == want.OM and np.array_equal(o, want.cols.ravel())
            cells = sum(mo.band_cells(p[2], p[3]) for p in pairs)
            q.put((ok, totals, (n, cells, 0)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world,n", [(2, 23), (3, 10), (2, 1)])
def test_scatter_compute_gather(world, n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, totals, want = q.get(timeout=180)
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
        loads = np.array([cost[p].sum() for p in parts])
        assert loads.max() - loads.min() <= cost.max()          # LPT bound
    batch, _ = _batch(3, 12)
    sub = shard.take(batch, np.array([7, 2, 9]))
    assert np.array_equal(shard.pair_cost(sub), shard.pair_cost(batch)[[7, 2, 9]])
