"""The tree driver's host side -- list walks in pieces, snapshots that share text with the blocks they were taken of, retired blocks freed
in the background, rows in one allocation, projection of long lists on arrays, replay into lists -- under AddressSanitizer +
UndefinedBehaviorSanitizer on the CPU, over the STAND-IN aligner of tests/tools/hostprof (every merge answered with the two slices side by
side: the blocks' shapes, not an alignment -- the GPU tests compare the real thing with the stock programs).  What is checked: no report,
and the destination byte for byte the same whichever way the lists are cut and the blocks are freed."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tests", "tools", "hostprof")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", MZ_SAMPLER="0")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("hostprof")
    p = subprocess.run(["bash", os.path.join(TOOLS, "build.sh"), str(d / "bin"), "-DMZ_STAGE_THREADS=4", "-fsanitize=address,undefined",
                        "-fno-omit-frame-pointer", "-fno-pie"], capture_output=True, timeout=900)
    if p.returncode != 0:
        pytest.skip("sanitizer build of the host side failed: " + p.stderr.decode()[-500:])
    q = subprocess.run([sys.executable, os.path.join(TOOLS, "make_inputs.py"), str(d / "in"), "9", "150"], capture_output=True, timeout=600)
    assert q.returncode == 0, q.stderr.decode()[-1000:]
    return str(d / "bin" / "roast_prof"), str(d / "in"), q.stdout.decode().strip()


def _run(harness, extra_env, out):
    exe, indir, tree = harness
    files = sorted(f for f in os.listdir(indir) if f.endswith(".sing.maf"))
    p = subprocess.run([exe, "E=ref", tree] + files + [out], cwd=indir, capture_output=True, timeout=600, env=dict(ENV, **extra_env))
    err = p.stderr.decode()
    assert p.returncode == 0 and "Sanitizer" not in err and "runtime error" not in err, err[-3000:]
    body = [l for l in open(os.path.join(indir, out)).read().split("\n") if not l.startswith("#")]
    assert sum(l.startswith("a score=") for l in body) > 200
    return body


def test_tree_driver_host_side_under_the_sanitizers(harness):
    want = _run(harness, {}, "plain.maf")
    # lists cut into pieces of a few blocks, long-list projection on short lists
    assert _run(harness, {"MZ_WALK_PIECE_MIN": "8", "MZ_FUSE_PARALLEL_MIN": "40"}, "pieces.maf") == want
    # retired blocks freed on the spot instead of in the background
    assert _run(harness, {"MZ_REAPER": "0", "MZ_WALK_PIECE_MIN": "8"}, "inline.maf") == want
    # the nodes above one leaf run on MAF text, fed by lists from below
    assert _run(harness, {"MZ_ROAST_TAINT": "sab"}, "taint.maf") == want
    # text everywhere
    assert _run(harness, {"MZ_ROAST_TEXT": "1"}, "text.maf") == want


def test_multiz_command_line_over_the_stand_in(harness):
    exe, indir, _ = harness
    files = sorted(f for f in os.listdir(indir) if f.endswith(".sing.maf"))[:2]
    outs = []
    for k, env in enumerate(({}, {"MZ_WALK_PIECE_MIN": "6"}, {"MZ_REAPER": "0"})):
        d = os.path.join(indir, "mz%d" % k)
        os.makedirs(d)
        p = subprocess.run([exe, "M=30", "../" + files[0], "../" + files[1], "1", "u1", "u2"], cwd=d, capture_output=True, timeout=300,
                           env=dict(ENV, HOSTPROF_MAIN="multiz", **env))
        err = p.stderr.decode()
        assert p.returncode == 0 and "Sanitizer" not in err and "runtime error" not in err, err[-3000:]
        outs.append((p.stdout, open(os.path.join(d, "u1"), "rb").read(), open(os.path.join(d, "u2"), "rb").read()))
    assert outs[0][0].count(b"a score=") > 20
    assert outs[1] == outs[0] and outs[2] == outs[0]
