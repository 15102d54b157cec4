"""The exchange of multiz_amd/csrc/mz_shard.c (include/mz_shard.h) on CPU: a list that exists on one rank is dealt out, every share's
link image travels, result images come back and the root assembles the merged columns from its own A and B -- the C code itself, over
its loop-back transport (every rank in this process: one after the other, and as threads) and over a caller-supplied one.  Where the
product aligns a share on its GPU (mz_shard_align) the oracle fills in the result image (tests/linkfmt.py)."""
import threading

import numpy as np
import pytest

import linkfmt
from multiz_amd import api
from oracle import mzoracle as mo
from test_shard_gloo import _batch


def _align_with_oracle(sh):
    image, exc = sh.host_image()
    res, cells, failed = linkfmt.oracle_result_image(sh.desc, image, exc)
    sh.set_result(res)
    return cells, failed


def _check_root(outs, failed, batch, pairs, index_sets):
    assert failed == 0 and (outs["status"] == 0).all()
    W = batch["K"].astype(np.int64) + batch["L"]
    for i, (A, B, LB, RB) in enumerate(pairs):
        want = mo.yama(A, B, LB, RB)
        got = np.ctypeslib.as_array((api.C.c_uint8 * (int(outs["OM"][i]) * int(W[i]))).from_address(int(outs["cols"][i])))
        assert int(outs["OM"][i]) == want.OM and np.array_equal(got, want.cols.ravel()), i
    allidx = np.sort(np.concatenate(index_sets))
    assert np.array_equal(allidx, np.arange(len(pairs)))          # every pair on exactly one rank


@pytest.mark.parametrize("world,n", [(3, 23), (2, 1), (1, 9), (4, 3)])
def test_loopback_ranks_one_after_the_other(world, n):
    batch, pairs = _batch(11, n)
    jobs, _ = api.host_jobs(batch)
    comms = api.Comm.loopback(world)
    s0, r0 = api.shard_traffic()
    root = world - 1 if world > 2 else 0
    order = [root] + [r for r in range(world) if r != root]      # the root's sends are in the mailbox when the others come to receive
    shards = {}
    for r in order:
        shards[r] = api.Shard(comms[r], root, jobs if r == root else None)
    assert sum(sh.n for sh in shards.values()) == n
    for sh in shards.values():
        _align_with_oracle(sh)
    outs = None
    for r in order[1:] + [root]:                                  # ... and theirs when the root comes to gather
        o, failed = shards[r].gather()
        if r == root:
            outs = o
            _check_root(outs, failed, batch, pairs, [s.index for s in shards.values()])
    s1, r1 = api.shard_traffic()
    assert world == 1 or (s1 > s0 and r1 > r0 and s1 - s0 == r1 - r0)      # what was sent arrived, byte for byte
    api.free_outs(outs)
    for sh in shards.values():
        sh.free()
    for c in comms:
        c.free()


def test_loopback_ranks_as_threads():
    world, n, root = 3, 40, 1
    batch, pairs = _batch(5, n)
    jobs, _ = api.host_jobs(batch)
    comms = api.Comm.loopback(world)
    result, errors, idx = {}, [], [None] * world

    def rank(r):
        try:
            sh = api.Shard(comms[r], root, jobs if r == root else None)
            idx[r] = sh.index
            _align_with_oracle(sh)
            o, failed = sh.gather()
            if r == root:
                result["outs"], result["failed"] = o, failed
            sh.free()
        except Exception as e:                                    # noqa: BLE001
            errors.append((r, repr(e)))
    ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errors, errors
    _check_root(result["outs"], result["failed"], batch, pairs, idx)
    api.free_outs(result["outs"])
    for c in comms:
        c.free()


def test_the_callers_own_transport_and_a_share_nobody_aligned():
    """mz_comm_custom: two ranks whose send / recv are Python functions over queues (what multiz_amd/shard.py does with
    torch.distributed where there is no RCCL); a gather before the share has a result is refused"""
    import queue
    world, n, root = 2, 12, 0
    batch, pairs = _batch(2, n)
    jobs, _ = api.host_jobs(batch)
    box = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
    comms = []
    for r in range(world):
        def send(buf, peer, r=r):
            box[(r, peer)].put(buf.copy())

        def recv(buf, peer, r=r):
            m = box[(peer, r)].get(timeout=30)
            assert m.size == buf.size
            buf[:] = m
        comms.append(api.Comm.custom(r, world, send, recv))
    shards = [api.Shard(comms[0], root, jobs), api.Shard(comms[1], root, None)]
    with pytest.raises(RuntimeError, match="has not been aligned"):
        shards[1].gather()
    for sh in shards:
        _align_with_oracle(sh)
    shards[1].gather()
    outs, failed = shards[0].gather()
    _check_root(outs, failed, batch, pairs, [s.index for s in shards])
    api.free_outs(outs)
    for sh in shards:
        sh.free()
    for c in comms:
        c.free()


def test_a_result_image_of_another_list_is_refused():
    batch, pairs = _batch(7, 6)
    jobs, _ = api.host_jobs(batch)
    comms = api.Comm.loopback(1)
    sh = api.Shard(comms[0], 0, jobs)
    image, exc = sh.host_image()
    res, _, _ = linkfmt.oracle_result_image(sh.desc, image, exc)
    for short in (res[: 64 + 8], res[:0]):                       # too short for six records; no image at all (read as "trusted" once)
        with pytest.raises(RuntimeError, match="cannot be the result image"):
            sh.set_result(short)
    bad = res.copy()
    bad[64: 64 + 40 * 6].view(api.RES_DT)["off"][2] = 1 << 40    # records that point outside the image
    sh.set_result(bad)
    with pytest.raises(RuntimeError, match="does not belong"):
        sh.gather()
    sh.free()
    comms[0].free()


def test_a_peer_that_answers_with_no_image_is_refused():
    """the root checks what a PEER says about its result image before it reads any of it: a header of [pairs, 0 bytes] used to be taken
    for "the pipeline's own image" and its records read from a one-byte block"""
    batch, pairs = _batch(5, 8)
    jobs, _ = api.host_jobs(batch)
    state = {"share": None}

    def send(buf, peer):
        pass                                                     # (whatever the root scatters is dropped: the peer is this test)

    def recv(buf, peer):
        if buf.size == 16:                                       # the gather's header: [pairs of the share, bytes of its result image]
            buf.view(np.int64)[:] = [state["share"], 0]
        else:
            buf[:] = 0

    comm = api.Comm.custom(0, 2, send, recv)
    sh = api.Shard(comm, 0, jobs)
    state["share"] = len(jobs) - sh.n
    _align_with_oracle(sh)
    with pytest.raises(RuntimeError, match="result image of 0 bytes"):
        sh.gather()
    sh.free()
    comm.free()


def test_more_ranks_than_gpus_fit_a_node():
    """the dealing rule is shared with mz_yama_batch()'s GPUs of one process (at most 16); ranks are any number (a world of 17
    overran a stack array)"""
    world, n = 40, 53
    batch, pairs = _batch(3, n)
    jobs, _ = api.host_jobs(batch)
    comms = api.Comm.loopback(world)
    shards = [api.Shard(comms[0], 0, jobs)] + [api.Shard(comms[r], 0, None) for r in range(1, world)]
    assert sum(sh.n for sh in shards) == n and max(sh.n for sh in shards) <= 2
    for sh in shards:
        _align_with_oracle(sh)
    for sh in shards[1:]:
        sh.gather()
    outs, failed = shards[0].gather()
    _check_root(outs, failed, batch, pairs, [s.index for s in shards])
    api.free_outs(outs)
    for sh in shards:
        sh.free()
    for c in comms:
        c.free()


# ---------------------------------------------------------------------------------------------------------------- the exchange in chunks
# mz_shard_run(): packing, moving, aligning and assembling of different chunks side by side (include/mz_shard.h).  Every rank is a
# thread here (the steps of different ranks wait for each other's messages); the oracle stands in for the GPU through the align hook.

def _oracle_align(chunk, desc, image, exc):
    return linkfmt.oracle_result_image(desc, image, exc)[0]


def _run_as_threads(world, root, jobs, chunks, align_of=lambda r: _oracle_align):
    comms = api.Comm.loopback(world)
    out, errors = {}, {}

    def rank(r):
        try:
            out[r] = api.shard_run(comms[r], root, jobs if r == root else None, chunks=chunks, align=align_of(r))
        except Exception as e:                                    # noqa: BLE001
            errors[r] = e
    ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    for c in comms:
        c.free()
    return out, errors


@pytest.mark.parametrize("world,n,chunks,root", [(3, 41, 4, 1), (2, 1, 3, 0), (1, 9, 2, 0), (4, 3, 2, 3), (2, 30, 1, 0), (20, 50, 3, 0), (3, 0, 2, 0)])
def test_chunked_exchange_ranks_as_threads(world, n, chunks, root):
    """every (rank, chunk) bin gets the same mix or nothing: with fewer pairs than bins most chunks are EMPTY (a header, no image, no
    result); 20 ranks: more than one process has GPUs; one chunk: the pipeline's shortest form; no pairs at all"""
    batch, pairs = _batch(5, n) if n else (None, [])
    jobs = api.host_jobs(batch)[0] if n else np.zeros(0, dtype=api.JOB_DT)
    s0, r0 = api.shard_traffic()
    out, errors = _run_as_threads(world, root, jobs, chunks)
    assert not errors, errors
    outs, failed, times = out[root]
    assert failed == 0 and times["chunks"] == chunks and times["steps"] == chunks + 4
    assert sum(out[r][2]["pairs"] for r in range(world)) == n                 # every pair was aligned on exactly one rank
    assert sum(out[r][2]["cells"] for r in range(world)) == sum(mo.band_cells(p[2], p[3]) for p in pairs)
    if n:
        assert max(out[r][2]["pairs"] for r in range(world)) <= -(-n // world) + chunks       # the snake's balance, chunk by chunk
        _check_root(outs, failed, batch, pairs, [np.arange(n)])
        api.free_outs(outs)
    s1, r1 = api.shard_traffic()
    assert world == 1 or n == 0 or (s1 - s0 == r1 - r0 and s1 > s0)              # what was sent arrived, byte for byte
    assert all(out[r][0] is None for r in range(world) if r != root)


def test_chunked_exchange_a_chunk_that_fails_on_one_rank():
    """a rank whose align fails for one chunk says so in that chunk's result header: the root leaves exactly those pairs without a result
    (MZ_E_DEVICE), everybody's exchange runs to its end, and the rank that failed reports its error"""
    world, n, chunks, root = 3, 36, 3, 0
    batch, pairs = _batch(9, n)
    jobs, _ = api.host_jobs(batch)
    seen = []

    def bad_align(chunk, desc, image, exc):
        seen.append((chunk, int(desc[0])))
        if chunk == 1:
            raise ValueError("this GPU is on fire")
        return _oracle_align(chunk, desc, image, exc)
    out, errors = _run_as_threads(world, root, jobs, chunks, align_of=lambda r: bad_align if r == 2 else _oracle_align)
    assert list(errors) == [2] and "on fire" in repr(errors[2])
    outs, failed, times = out[root]
    lost = dict(seen)[1]
    assert failed == lost > 0 and int((outs["status"] != 0).sum()) == lost
    W = batch["K"].astype(np.int64) + batch["L"]
    for i, (A, B, LB, RB) in enumerate(pairs):
        if outs["status"][i] == 0:
            want = mo.yama(A, B, LB, RB)
            got = np.ctypeslib.as_array((api.C.c_uint8 * (int(outs["OM"][i]) * int(W[i]))).from_address(int(outs["cols"][i])))
            assert int(outs["OM"][i]) == want.OM and np.array_equal(got, want.cols.ravel()), i
        else:
            assert api.MZ_STATUS[int(outs["status"][i])] == "device" and outs["cols"][i] == 0
    api.free_outs(outs)


def test_chunked_exchange_refuses_a_peers_short_result_image():
    world, n, chunks = 2, 12, 2
    batch, pairs = _batch(4, n)
    jobs, _ = api.host_jobs(batch)

    def short_align(chunk, desc, image, exc):
        return _oracle_align(chunk, desc, image, exc)[: 64 + 8]
    out, errors = _run_as_threads(world, 0, jobs, chunks, align_of=lambda r: short_align if r == 1 else _oracle_align)
    # the peer's own library refuses to take the image from its align (and sends the chunk as failed); nothing short reaches the root
    assert list(errors) == [1] and "cannot be the result image" in repr(errors[1])
    outs, failed, _ = out[0]
    assert failed == int((outs["status"] != 0).sum()) > 0
    api.free_outs(outs)
