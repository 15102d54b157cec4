"""Every BASELINE.json configuration at its FULL size on the HIP path, every pair against the CPU oracle (VERDICT r2,
item 4: round 2 tested C3 at 1 000 of 5 000 pairs, C4 at 20 000 of one GPU's 125 000, C5 at 64 of 1 000, C1 at 30 blocks):

    C1  two MAFs x 100 blocks x ~200 columns, reference row + one species row each, through the drivers
    C3  5 000 pairs, 10+10 rows, ~2k x 2k
    C4  125 000 pairs (one GPU's share of the 1M-pair 30-way tree workload)
    C5  1 000 pairs of ~100k x 100k, band radius 30

C3-C5 run BOTH product paths -- the host-buffer path mz_yama_batch() (class nibbles up, edit scripts back, columns
assembled on the host) and the device-resident path bench.py times -- and compare the per-pair hash of (OM, merged
column bytes) with the oracle's integer-exact profile restatement on all pairs, plus the compiled reference itself
(oracle/_ref/libref.so, where it travelled) on a seeded sample.  (C2 at full size is bench.py's own parity gate.)"""
import os
import subprocess

import numpy as np
import pytest

import inputs
from oracle import mzoracle as mo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def cpu_threads() -> int:
    """the container's CPU budget (affinity capped by the cgroup quota): more threads than that only get throttled"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


@pytest.fixture(scope="module")
def mz():
    import multiz_amd as m
    m.api.init(0)
    m.lib().mz_enable_fast(1)
    m.lib().mz_enable_row(1)
    return m


def _full_config(mz, cfg, first_pair=0, ref_sample=600):
    from multiz_amd import api, synth
    c = synth.CONFIGS[cfg]
    n = c["pairs"]
    batch = synth.make_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], first_pair=first_pair, indel=c.get("indel", 0))
    W = (batch["K"] + batch["L"]).astype(np.int32)
    # ---- the oracle: all pairs (profile form, integer-exact, pinned to the reference by tests/test_oracle_*.py)
    om, hs, cells, bad = mo.yama_batch(batch, variant=1, threads=cpu_threads())
    assert bad == 0
    # ---- the compiled reference itself on a sample
    if mo.have_reference():
        rng = np.random.default_rng(5)
        idx = np.sort(rng.choice(n, size=min(n, ref_sample), replace=False))
        r_om, r_hs, _, r_bad = mo.ref_batch(synth.subset(batch, idx), threads=cpu_threads())
        assert r_bad == 0 and np.array_equal(r_om, om[idx]) and np.array_equal(r_hs, hs[idx])
    # ---- host-buffer path
    jobs, outs = api.host_jobs(batch)
    assert api.yama_batch_records(jobs, outs) == 0
    assert (outs["status"] == 0).all() and np.array_equal(outs["OM"], om)
    got = mo.hash_cols(outs["cols"], outs["OM"], W)
    api.free_outs(outs)
    assert int((got != hs).sum()) == 0, f"host path: {int((got != hs).sum())} of {n} pairs differ"
    # ---- device-resident path (serial form, then the pipelined form on a second workspace)
    db = mz.DevBatch(batch)
    for form in ("serial", "pipelined"):
        w = db if form == "serial" else db.alternate()
        if form == "serial":
            w.run()
        else:
            w.run_async(); w.wait()
        res = w.results()
        assert (res["status"] == 0).all() and np.array_equal(res["om"], om) and int(res["cells"].sum()) == cells, form
        base = w.out.data_ptr()
        # the merged columns where they lie in HBM are hashed from a host copy
        host_out = w.out.cpu().numpy()
        ptr = host_out.ctypes.data + res["offOut"].astype(np.uint64)
        got = mo.hash_cols(ptr, res["om"], W)
        assert int((got != hs).sum()) == 0, f"{form}: {int((got != hs).sum())} of {n} pairs differ"
        del host_out, base
    return batch, om


def test_c3_full_size_5000_pairs(mz):
    batch, om = _full_config(mz, "c3")
    assert len(om) == 5000 and int(batch["K"][0]) == 10 and int(batch["L"][0]) == 10


def test_c4_one_gpu_share_125000_pairs(mz):
    batch, om = _full_config(mz, "c4", first_pair=2 * 125000, ref_sample=300)
    assert len(om) == 125000 and batch["K"].max() == 29 and (batch["K"] + batch["L"]).max() == 30


def test_c5_full_size_1000_pairs(mz):
    batch, om = _full_config(mz, "c5", ref_sample=48)
    assert len(om) == 1000 and batch["M"].min() >= 95000 and batch["M"].max() <= 105000


@pytest.mark.parametrize("cfg", ["c2i", "c2g", "c2w", "c2s", "c4i"])
def test_builder_configs_at_their_benched_size(mz, cfg):
    """bench.py's own configurations (not BASELINE's: C2 / C4 shapes whose bands come from blocks with indels, with a heavy tail, at
    radius 50 and 100 -- the lagged row kernel, the tagged wavefront, the strips, mixes of all of them in one batch) at the size
    their bench lines are quoted on, every pair against the oracle on both product paths (under -m gpu these
    kernels were only covered at 1 500 pairs and by the random sweeps)."""
    from multiz_amd import synth
    batch, om = _full_config(mz, cfg, ref_sample=200)
    assert len(om) == synth.CONFIGS[cfg]["pairs"]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "multiz_ref")), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("v", [1, 0])
def test_c1_two_100_block_mafs_literal(tmp_path, v):
    # BASELINE configs[0]: two MAFs of 100 blocks x ~200 columns, the reference row and one species row per block
    # (with the reference row alone nothing would reach yama()), merged by the stock binary on the CPU,
    # by the reference's own driver linked against libmzamd.so, and by the batched driver mz_multiz: identical bytes
    # on stdout and in both leftover files
    rng = np.random.default_rng(100 + v)
    ref = inputs.ACGT[rng.integers(0, 4, size=100 * 260 + 300)]
    f1, f2 = str(tmp_path / "a.maf"), str(tmp_path / "b.maf")
    b1 = inputs.random_maf_file(rng, ref, 100, 2, "p", stride=260, blen=(180, 230))
    b2 = inputs.random_maf_file(rng, ref, 100, 2, "q", stride=260, blen=(180, 230))
    assert len(b1) == 100 and len(b2) == 100 and all(len(b.rows) == 2 for b in b1 + b2)
    inputs.write_maf(f1, b1); inputs.write_maf(f2, b2)

    def run(binary, tag):
        d = tmp_path / tag
        d.mkdir()
        p = subprocess.run([binary, "../a.maf", "../b.maf", str(v), "u1", "u2"], capture_output=True, timeout=600, cwd=str(d))
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        return p.stdout, (d / "u1").read_bytes(), (d / "u2").read_bytes()

    want = run(os.path.join(REF, "multiz_ref"), "ref")
    assert want[0].count(b"a score=") >= 60                  # most of the 100 + 100 blocks overlap and merge
    assert run(os.path.join(REF, "multiz_mzamd"), "dropin") == want
    assert run(os.path.join(ROOT, "multiz_amd", "mz_multiz"), "batched") == want


@pytest.mark.parametrize("cfg,v", [("c2", 1), ("c2", 0), ("c4", 1)])
def test_text_path_full_size(mz, cfg, v):
    """mz_preyama_batch() -- block TEXT in, block rows out: what mz_multiz / mz_roast run every merge through -- at a BASELINE
    configuration's full size (C2: 50 000 merges, one-stage and two-stage; C4: one GPU's 125 000).  Every merge: the status, and what
    no aligner may change -- a merged row without its dashes IS its source row without its dashes, in the order mafBuild() keeps
    (all of block 1, then block 2 below its reference row), the base counts are the rows' own, every row is OM columns long, no
    merged column is all dashes.  A seeded sample of 1 000 merges against the compiled reference's pre_yama(): rows, counts, score."""
    import bench
    from multiz_amd import api, synth
    c = synth.CONFIGS[cfg]
    n = c["pairs"]
    pb = synth.make_pre_batch(n, c["K"], c["L"], c["mlo"], c["mhi"], c["radius"], events=c.get("indel", 0), v=v)
    jobs, outs = pb["jobs"], pb["outs"]
    api.preyama_batch_records(jobs, outs)
    try:
        assert (outs["status"] == 0).all()
        if v == 0:
            assert int((outs["stage"] == 2).sum()) >= n // 2            # the two-stage path really ran
        step = 1 if n <= 50000 else 3                                  # (Python loop: every merge of C2, every third of C4's 125 000)
        for i in range(0, n, step):
            r1, r2 = synth.pre_rows_of(pb, i)
            src = r1 + r2[1:]
            rows, size = api.preout_rows(outs, i, len(src))
            if rows is None:
                assert int(outs["null_result"][i]) != 0, i
                continue
            om = int(outs["OM"][i])
            cols_alive = np.zeros(om, dtype=bool)
            for k, (t, s) in enumerate(zip(rows, src)):
                assert len(t) == om
                bases = t.replace(b"-", b"")
                assert bases == s.replace(b"-", b""), (i, k)
                assert int(size[k]) == len(bases), (i, k)
                cols_alive |= np.frombuffer(t, dtype=np.uint8) != 0x2D
            assert cols_alive.all(), i
        idx = np.sort(np.random.default_rng(9).choice(n, size=1000, replace=False))
        bad, what = bench.pre_check(pb, outs, idx, c["radius"])
        assert bad == 0, f"{bad} of 1000 sampled merges differ from the {what}"
    finally:
        api.free_preouts(outs)
