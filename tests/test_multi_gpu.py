"""Several GPUs in one process (include/mz_amd.h: mz_init_multi, MZ_NGPU) and the N > 1 forms of bench.py, as far as
a one-GPU box can exercise them: the context table, the dealing of a host batch over contexts with one host thread
each (two contexts on GPU 0, MZ_ALLOW_DUP_DEVICES=1), MZ_NGPU from the environment, and bench.py's rank spawning,
weak-scaling and --scatter (RCCL-style scatter / device compute / gather; gloo here, both ranks on GPU 0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, ctypes as C, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import multiz_amd as mz
from multiz_amd import api, synth
from oracle import mzoracle as mo
lib = mz.lib()
lib.mz_init_multi.argtypes = [C.c_int, C.POINTER(C.c_int)]
mode = sys.argv[1]
if mode == "dup":
    devs = (C.c_int * 2)(0, 0)
    assert lib.mz_init_multi(2, devs) == 0, lib.mz_last_error()
    assert lib.mz_device_count() == 2
elif mode == "env":
    pass                                            # MZ_NGPU=1 MZ_DEVICE=0 from the environment, on first use
elif mode == "bad":
    devs = (C.c_int * 2)(0, 0)
    assert lib.mz_init_multi(2, devs) == -1 and b"listed twice" in lib.mz_last_error()
    assert lib.mz_init_multi(64, None) == -1
    devs = (C.c_int * 2)(0, 7)
    import torch
    if torch.cuda.device_count() < 8:
        assert lib.mz_init_multi(2, devs) == -1 and b"out of range" in lib.mz_last_error()
    assert lib.mz_device_count() == 0
    print("multi ok"); sys.exit(0)
n = 9000
batch = synth.make_batch(n, 0, 0, 200, 600, 30, first_pair=11)          # tree workload: pairs of very different cost
jobs, outs = api.host_jobs(batch)
for rep in range(2):
    assert api.yama_batch_records(jobs, outs) == 0
    om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=16)
    assert bad == 0 and (outs["status"] == 0).all() and np.array_equal(outs["OM"], om)
    W = batch["K"].astype(np.int64) + batch["L"]
    for i in range(0, n, 7):
        got = np.frombuffer(C.string_at(int(outs["cols"][i]), int(om[i]) * int(W[i])), dtype=np.uint8)
        assert mo.fnv1a_np(got, mo.fnv1a_np(np.array([om[i]], dtype=np.int32).view(np.uint8))) == int(hs[i]), i
    api.free_outs(outs)
assert lib.mz_device_count() == (2 if mode == "dup" else 1)
# bands of blocks with indels: row-parallel, lagged and wavefront pairs in every chunk, their DP kernels side by side -- on
# both contexts at once in "dup" mode (one device, two host threads: the device's side streams are shared)
batch = synth.make_batch(n, 2, 2, 300, 600, 30, first_pair=3, indel=10)
jobs, outs = api.host_jobs(batch)
assert api.yama_batch_records(jobs, outs) == 0
om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=16)
assert bad == 0 and (outs["status"] == 0).all() and np.array_equal(outs["OM"], om)
for i in range(0, n, 3):
    got = np.frombuffer(C.string_at(int(outs["cols"][i]), int(om[i]) * 4), dtype=np.uint8)
    assert mo.fnv1a_np(got, mo.fnv1a_np(np.array([om[i]], dtype=np.int32).view(np.uint8))) == int(hs[i]), i
api.free_outs(outs)
if mode == "dup":
    # block text in / block text out over both contexts (mz_preyama_batch): 60 distinct block pairs, 4 800 jobs
    import inputs
    from test_preyama import _prejob, _assemble, same_block
    rng = np.random.default_rng(3)
    base = []
    while len(base) < 60:
        a1, a2, beg, end = inputs.random_block_pair(rng, int(rng.integers(1, 5)), int(rng.integers(2, 5)), int(rng.integers(80, 400)))
        if end - beg < 12:
            continue
        try:
            want, _ = mo.pre_yama(a1, a2, beg, end, 30, 1)
        except RuntimeError:
            continue
        j, cb1, cb2 = _prejob(a1, a2, beg, end, 30)
        base.append((j, a1, cb1, a2, cb2, want))
    jobs = [base[i % 60][0] for i in range(4800)]
    res = mz.preyama_batch(jobs)
    for i, r in enumerate(res):
        j, a1, cb1, a2, cb2, want = base[i % 60]
        assert r["status"] == 0 and same_block(_assemble(r, a1, cb1, a2, cb2), want), i
print("multi ok")
'''


@pytest.mark.parametrize("mode,env", [("dup", {"MZ_ALLOW_DUP_DEVICES": "1"}), ("env", {"MZ_NGPU": "1", "MZ_DEVICE": "0"}), ("bad", {})])
def test_contexts_and_dealing(mode, env):
    p = subprocess.run([sys.executable, "-c", CODE, mode], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, timeout=900)
    assert p.returncode == 0 and b"multi ok" in p.stdout, p.stderr.decode()[-3000:]


@pytest.mark.parametrize("extra", [[], ["--scatter"]])
def test_bench_two_ranks_on_one_gpu(extra):
    # `--gpus 2` without a distributed environment: bench.py starts the ranks itself; MZ_BENCH_SHARE_GPU=1 puts both on
    # GPU 0 over gloo (development switch; the number is not a scaling measurement, the control flow is the real one)
    env = dict(os.environ, MZ_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "c4", "--pairs", "3000", "--steps", "3", "--warmup", "1"] + extra,
                       cwd=ROOT, env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().split("\n") if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["pairs_total"] == 6000 and d["value"] > 0
    # the headline is the transfer-inclusive rate of K = --steps calls of mz_yama_batch() on every rank; the resident pipeline's rate beside it
    assert d["steps"] == 3 and len(d["host_ms_all"]) == 3 and d["value_resident"] > d["value"] and "mz_yama_batch" in d["value_is"]
    # the self-verifying part of a multi-GPU line: which device every rank held, every rank's own rate, the spread
    assert len(d["devices"]) == 2 and d["distinct_devices"] == 1 and "gloo" in d["backend"]      # (both ranks on GPU 0 here)
    assert len(d["per_rank_gcups"]) == 2 and min(d["per_rank_gcups"]) > 0 and d["rank_max_over_min"] >= 1.0
    if extra:
        ex = d["exchange"]
        assert ex["ranks_used"] == 2 and "RCCL" in d["config"]["parallelism"] and ex["format"].startswith("link images")
        bp = ex["bytes_per_pair"]                                # class nibbles + band steps out, records + 2-bit scripts back
        assert bp["out"] < 0.5 * bp["pools_out"] and bp["out"] <= 8192 and bp["back"] <= 1024 and bp["back"] < 0.1 * bp["columns_back"]
        assert ex["checked_pairs"] >= 400                        # the root's assembled columns against the compiled reference


def test_bench_one_process_two_contexts():
    # `--mode ngpu`: the C path (mz_init_multi + mz_yama_batch over a host list) that mz_multiz / mz_roast use on a node;
    # two contexts on GPU 0 here (MZ_ALLOW_DUP_DEVICES=1)
    env = dict(os.environ, MZ_ALLOW_DUP_DEVICES="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--mode", "ngpu", "--gpus", "2", "--config", "c4", "--pairs", "6000", "--steps", "3", "--warmup", "1"],
                       cwd=ROOT, env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads([l for l in p.stdout.decode().split("\n") if l.startswith("{")][-1])
    assert d["mode"] == "ngpu" and d["n_gpus"] == 2 and d["config"]["pairs_total"] == 12000 and d["value"] > 0
    assert len(d["devices"]) == 2 and d["distinct_devices"] == 1 and d["link_bytes_per_pair"]["up"] > 0


def test_bench_refuses_a_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--steps", "1"], cwd=ROOT, env=env, capture_output=True, timeout=300)
    assert p.returncode != 0 and b"WORLD_SIZE=1" in p.stderr
