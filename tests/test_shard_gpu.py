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


def _link_worker(rank, world, port, n, cfg, q):
    """link images: nibbles + band steps to the rank, mz_link_plan / mz_link_finish where they land, records + 2-bit scripts back,
    merged columns assembled by the root from its own pools"""
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
            batch = synth.make_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=77, indel=c.get("indel", 0))
        sh, totals = shard.run_sharded_link(batch, lambda s: shard.link_compute(s, device="cuda:0"))
        if rank == 0:
            om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=8)
            ok = bad == 0 and bool((sh.status == 0).all()) and set(sh.owner.tolist()) == set(range(world))
            mism = 0
            for i in range(n):
                m_ = int(sh.om[i])
                h = mo.fnv1a_np(sh.cols(i), mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8)))
                mism += (m_ != om[i]) or (h != int(hs[i]))
            ex = dict(shard.last_exchange)
            pools = sum(batch[k].nbytes for k in ("poolA", "poolB", "poolLB", "poolRB"))
            merged = int((sh.om.astype(np.int64) * sh.widths).sum())
            ok &= ex["up_bytes"] < 0.6 * pools and ex["down_bytes"] < 0.25 * merged + 100 * n
            sh.release()
            q.put((ok, mism, totals, (n, cells, 0)))
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def _chunk_worker(rank, world, port, n, cfg, q):
    """the exchange in chunks (mz_shard_run) with the PRODUCT's align: every chunk's image goes up to the rank's GPU, mz_link_plan /
    mz_link_finish run on it while the transport moves the next chunk and the root packs the one after and assembles the one before"""
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
            batch = synth.make_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=77, indel=c.get("indel", 0))
        sh, totals, times = shard.run_sharded_chunks(batch, None, chunks=5)
        assert times["chunks"] == 5 and times["pairs"] > 0 and times["align_s"] > 0
        if rank == 0:
            om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=8)
            ok = bad == 0 and bool((sh.status == 0).all())
            mism = 0
            for i in range(n):
                m_ = int(sh.om[i])
                h = mo.fnv1a_np(sh.cols(i), mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8)))
                mism += (m_ != om[i]) or (h != int(hs[i]))
            sh.release()
            q.put((ok, mism, totals, (n, cells, 0)))
        torch.cuda.synchronize()
        shard.drop_comms()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("cfg,n,worker", [("c4", 600, "pools"), ("c2", 300, "pools"), ("c4", 3000, "link"), ("c2i", 1500, "link"),
                                          ("c4", 3000, "chunks"), ("c2i", 1500, "chunks")])
def test_scatter_device_compute_gather_two_ranks(cfg, n, worker):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target={"pools": _worker, "link": _link_worker, "chunks": _chunk_worker}[worker], args=(r, 2, port, n, cfg, q)) for r in range(2)]
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


def test_result_image_is_the_oracles():
    """one process: the product's result image for a link image (mz_link_plan / mz_link_finish) holds the records and the edit
    scripts the oracle's traceback gives (tests/linkfmt.py), refused pairs included; mz_link_expand gives the batch the image decodes to"""
    import torch
    import inputs
    import linkfmt
    import multiz_amd as mz
    from multiz_amd import api
    from oracle import mzoracle as mo
    from test_link_image import _jobs, _pairs
    mz.api.init(0)
    pairs = _pairs(21, 120)
    A, B, LB, RB = pairs[5]
    LB = LB.copy(); LB[3] = LB[2] - 1 if LB[2] > 0 else 0; RB = RB.copy(); RB[len(RB) // 2] = RB[len(RB) // 2 - 1] - 1     # a band the reference refuses
    pairs[5] = (A, B, LB, RB)
    jobs = _jobs(pairs)
    desc, image, exc = api.link_pack(jobs)
    want, cells, failed = linkfmt.oracle_result_image(desc, image, exc)
    t_im, t_ex = torch.from_numpy(image).cuda(), torch.from_numpy(exc).cuda()
    res = api.link_run(desc, t_im, t_ex)
    torch.cuda.synchronize()
    got = res.cpu().numpy()
    n = len(pairs)
    rg, rw = api.link_records(got, n), api.link_records(want, n)
    assert failed >= 1 and (rg["status"] != 0).sum() == failed
    for k in ("status", "om", "cells"):
        assert np.array_equal(rg[k], rw[k]), k
    okp = rg["status"] == 0
    assert np.array_equal(rg["f"][okp], rw["f"][okp]) and np.array_equal(rg["badrow"][~okp], rw["badrow"][~okp])
    sg, sw = 64 + linkfmt.al256(40 * n), 64 + linkfmt.al256(40 * n)
    for p in np.flatnonzero(okp):
        nb = (int(rg["om"][p]) + 3) // 4
        assert np.array_equal(got[sg + rg["off"][p]: sg + rg["off"][p] + nb], want[sw + rw["off"][p]: sw + rw["off"][p] + nb]), p
    from multiz_amd import shard
    tot = shard.link_totals(res, n)
    assert tot["cells"] == cells and tot["failed"] == failed
    # the expansion as a device-resident batch of its own
    t = api.link_expand(desc, t_im, t_ex)
    torch.cuda.synchronize()
    dec = linkfmt.decode_image(desc, image, exc)
    for k in ("K", "L", "M", "N", "offA", "offB", "offBand"):
        assert np.array_equal(t[k].cpu().numpy(), dec[k]), k
    assert np.array_equal(t["poolLB"].cpu().numpy()[: len(dec["poolLB"])], dec["poolLB"]) and np.array_equal(t["poolRB"].cpu().numpy()[: len(dec["poolRB"])], dec["poolRB"])
    for p in range(n):
        K, L, M, N = (int(dec[k][p]) for k in ("K", "L", "M", "N"))
        a0, b0 = int(dec["offA"][p]), int(dec["offB"][p])
        assert np.array_equal(t["poolA"][a0: a0 + K * M].cpu().numpy(), dec["poolA"][a0: a0 + K * M])
        assert np.array_equal(t["poolB"][b0: b0 + L * N].cpu().numpy(), dec["poolB"][b0: b0 + L * N])
    db = mz.DevBatch.from_tensors(t, device="cuda:0")
    db.run()
    r = db.results()
    assert np.array_equal(r["status"], rg["status"]) and np.array_equal(r["om"][okp], rg["om"][okp])
    outs = api.link_assemble(jobs, got)
    for p in np.flatnonzero(okp):
        A, B, LB, RB = pairs[p]
        w = mo.yama(A, B, LB, RB)
        c = np.ctypeslib.as_array((api.C.c_uint8 * (w.OM * (A.shape[1] + B.shape[1]))).from_address(int(outs["cols"][p])))
        assert outs["OM"][p] == w.OM and np.array_equal(c, w.cols.ravel()), p
    api.free_outs(outs)


RCCL_CODE = r'''
import sys, numpy as np
sys.path.insert(0, ".")
import multiz_amd as mz
from multiz_amd import api, synth
from oracle import mzoracle as mo
api.init(0)
comm = api.Comm.rccl(api.Comm.rccl_unique_id(), 0, 1)                 # librccl is loaded here, by the library
comm.echo(1 << 20)                                                     # ncclGroupStart; ncclSend + ncclRecv to this rank; ncclGroupEnd
comm.echo(4096 + 16)
c = synth.CONFIGS["c2i"]
n = 1200
batch = synth.make_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=5, indel=c.get("indel", 0))
jobs, _ = api.host_jobs(batch)
sh = api.Shard(comm, 0, jobs)                                           # mz_shard_scatter: the one rank's own share, in device memory
assert sh.n == n and sorted(sh.index.tolist()) == list(range(n))
sh.align()                                                             # mz_link_plan + mz_link_finish where the image lies
cells, failed = sh.totals()
outs, bad = sh.gather()                                                 # mz_shard_gather: assembled from the root's own A and B
om, hs, want_cells, refbad = mo.yama_batch(batch, variant=1, threads=8)
assert failed == 0 and bad == 0 and refbad == 0 and cells == want_cells, (failed, bad, cells, want_cells)
W = batch["K"].astype(np.int64) + batch["L"]
for i in range(n):
    m_ = int(outs["OM"][i])
    got = np.ctypeslib.as_array((api.C.c_uint8 * (m_ * int(W[i]))).from_address(int(outs["cols"][i])))
    assert m_ == om[i] and mo.fnv1a_np(got, mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8))) == int(hs[i]), i
api.free_outs(outs)
sh.free()
# ... and the same list through the exchange in chunks (mz_shard_run): six chunks, each aligned where the transport's device buffer lies
outs, bad, times = api.shard_run(comm, 0, jobs, chunks=6)
assert bad == 0 and times["chunks"] == 6 and times["pairs"] == n and times["cells"] == want_cells and times["failed"] == 0, times
for i in range(n):
    m_ = int(outs["OM"][i])
    got = np.ctypeslib.as_array((api.C.c_uint8 * (m_ * int(W[i]))).from_address(int(outs["cols"][i])))
    assert m_ == om[i] and mo.fnv1a_np(got, mo.fnv1a_np(np.array([m_], dtype=np.int32).view(np.uint8))) == int(hs[i]), i
api.free_outs(outs)
comm.free()
print("rccl ok", api.shard_traffic(), times)
'''


def test_rccl_world_of_one_through_the_c_leg():
    """The library's RCCL transport with the ranks the test box has -- one: the communicator is made, a grouped ncclSend / ncclRecv to
    itself moves a megabyte intact (mz_comm_echo), and scatter / align / gather run on it (no peer: the share does not travel, but
    the buffers are the transport's device buffers and the align runs where they lie).  A child process: RCCL keeps threads."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", RCCL_CODE], cwd=root, capture_output=True, timeout=900)
    assert p.returncode == 0 and b"rccl ok" in p.stdout, (p.stdout + p.stderr).decode()[-3000:]
