"""The host C of the product under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: `make -C
multiz_amd/csrc san`; GPU sanitizers are not available on this pool).  Host-only paths: the projection and tree parser
of mz_roast, the readers / list walks / unused-part printing of mz_multiz and mz_multic on inputs that need no merge,
and the band / packing / generator helpers through a small C program.  Any report fails the test."""
import os
import subprocess

import numpy as np
import pytest

import inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "multiz_amd", "san")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
           HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")


@pytest.fixture(scope="module")
def san():
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "multiz_amd", "csrc"), "san"], capture_output=True, timeout=900)
    if p.returncode != 0:
        pytest.skip("sanitizer build failed: " + p.stderr.decode()[-500:])
    return SAN


def _clean(p):
    err = p.stderr.decode()
    assert "Sanitizer" not in err and "runtime error" not in err, err[-3000:]


def test_projection_and_tree_parser(san, tmp_path):
    from test_roast_inprocess import _messy_maf
    for seed in (4, 5):
        path = str(tmp_path / f"in{seed}.maf")
        _messy_maf(np.random.default_rng(seed), path)
        p = subprocess.run([os.path.join(san, "mz_roast"), "--project", path, "ref"], capture_output=True, timeout=300, env=ENV)
        assert p.returncode == 0 and p.stdout.count(b"a score=") > 30
        _clean(p)
        q = subprocess.run([os.path.join(san, "mz_roast"), "--project", path, "ref"], capture_output=True, timeout=300,
                           env=dict(ENV, MZ_FUSE_PARALLEL_MIN="2"))              # (the fusion pass's long-list form)
        assert q.returncode == 0 and q.stdout == p.stdout
        _clean(q)
    p = subprocess.run([os.path.join(san, "mz_roast"), "-", "E=ref", "(((ref a) (b c)) (d (e f)))", "x", "o.maf"], capture_output=True, timeout=60, env=ENV)
    assert p.returncode == 0
    _clean(p)


def test_drivers_on_inputs_without_merges(san, tmp_path):
    rng = np.random.default_rng(11)
    ref = inputs.ACGT[rng.integers(0, 4, size=12 * 400 + 600)]
    b1 = inputs.random_maf_file(rng, ref, 12, 3, "p", stride=400, blen=(80, 150))
    b2 = inputs.random_maf_file(rng, ref[200:], 12, 2, "q", stride=400, blen=(80, 150))
    for b in b2:
        b.rows[0].start += 200
    inputs.write_maf(str(tmp_path / "a.maf"), b1)
    inputs.write_maf(str(tmp_path / "b.maf"), b2)
    for v in ("0", "1"):
        p = subprocess.run([os.path.join(san, "mz_multiz"), "a.maf", "b.maf", v, "u1", "u2"], cwd=str(tmp_path), capture_output=True, timeout=300, env=ENV)
        assert p.returncode == 0
        _clean(p)
    from test_batched_multic import overlapping_list
    inputs.write_maf(str(tmp_path / "c.maf"), overlapping_list(rng, ref, 10, 3, ("p", "p"), (260, 290)))
    inputs.write_maf(str(tmp_path / "d.maf"), overlapping_list(rng, ref, 10, 3, ("p", "p"), (300, 270)))
    p = subprocess.run([os.path.join(san, "mz_multic"), "c.maf", "d.maf", "1", "u1", "u2"], cwd=str(tmp_path), capture_output=True, timeout=300, env=ENV)
    assert p.returncode == 0
    _clean(p)


def test_host_helpers(san, tmp_path):
    exe = str(tmp_path / "san_host")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", os.path.join(ROOT, "tests", "tools", "san_host.c"), "-o", exe,
                           "-L" + san, "-lmzamd", "-Wl,-rpath," + san, "-Wl,-rpath,/opt/rocm/lib"])
    p = subprocess.run([exe], capture_output=True, timeout=300, env=ENV)
    assert p.returncode == 0 and b"san host ok" in p.stdout, p.stderr.decode()[-3000:]
    _clean(p)


def test_host_thread_pool_under_tsan_and_asan(tmp_path):
    """mz_pool.c alone (no GPU): loops waited for and loops posted, from several threads at once, a caller working in the pool, and a
    posted loop one piece of which dawdles -- it must be complete long before the dawdler is back (the piece is run a second time),
    and quiet only after."""
    src = [os.path.join(ROOT, "tests", "tools", "pool_stress.c"), os.path.join(ROOT, "multiz_amd", "csrc", "mz_pool.c")]
    for flags, name in ((["-fsanitize=thread"], "tsan"), (["-fsanitize=address,undefined"], "asan")):
        exe = str(tmp_path / ("pool_" + name))
        subprocess.check_call(["gcc", "-O1", "-g", "-I/opt/rocm/include"] + flags + src + ["-o", exe, "-lpthread"])
        # as shipped; then with every piece late at once, its second run starting late, and more second runs at a time than the default
        for extra in ({}, {"MZ_HEDGE_US": "1", "MZ_HEDGE_DELAY_US": "100", "MZ_HEDGE_DUPS": "6"}, {"MZ_HEDGE_FIRST": "0"}):
            p = subprocess.run([exe], capture_output=True, timeout=600, env=dict(ENV, MZ_HOST_THREADS="8", **extra))
            assert p.returncode == 0 and b"pool ok" in p.stdout, (p.stdout + p.stderr).decode()[-3000:]
            assert b"ThreadSanitizer" not in p.stderr
            _clean(p)
