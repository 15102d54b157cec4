"""The N > 1 path with the PRODUCT's compute: scatter -> multiz_amd.shard.device_compute (a DevBatch over the received
tensors, the HIP kernels through the C ABI) -> gather, two ranks.  The test box has one GPU, so both ranks use GPU 0
and the exchange runs over gloo (RCCL refuses two ranks on one device); on a multi-GPU node the same code runs with
backend "nccl" = RCCL and the tensors never leave the devices (bench.py --scatter)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n, cfg, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import multiz_amd as mz
        from multiz_amd import shard, synth
        from oracle import mzoracle as mo
        mz.api.init(0)
        batch = None
        if rank == 0:
            c = synth.CONFIGS[cfg]
            batch = synth.make_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=77)
        keep = []
        sh, totals = shard.run_sharded(batch, lambda s: shard.device_compute(s, device="cuda:0", keep=keep))
        if rank == 0:
            om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=8)
            ok = bad == 0 and bool((sh.status == 0).all()) and set(sh.owner.tolist()) == set(range(world))
            mism = 0
            for i in range(n):
                m_ = int(sh.om[i])
                h = mo.fnv1a_np(sh.cols(i), mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8)))
                mism += (m_ != om[i]) or (h != int(hs[i]))
            q.put((ok, mism, totals, (n, cells, 0)))
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("cfg,n", [("c4", 600), ("c2", 300)])
def test_scatter_device_compute_gather_two_ranks(cfg, n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, cfg, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue as _q
    for _ in range(600):
        try:
            ok, mism, totals, want = q.get(timeout=1)
            break
        except _q.Empty:
            assert all(p.exitcode in (None, 0) for p in procs), "a rank died"
    else:
        raise AssertionError("timed out")
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ok and mism == 0, (ok, mism)
    assert totals == want
