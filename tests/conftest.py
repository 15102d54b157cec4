import os
import sys

import numpy as np
import pytest

# torch first: it carries its own copy of the HIP runtime, and libmzamd.so (linked against /opt/rocm's) must find that
# one already loaded -- two HIP runtimes in one process do not share a device.  (multiz_amd.api does the same; a test
# that loads the library with ctypes before anything imported torch would otherwise decide the order.)
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden():
    z = np.load(os.path.join(ROOT, "tests", "golden", "yama_golden.npz"))
    tags = [str(t) for t in z["tags"]]
    cases = []
    for i, t in enumerate(tags):
        cases.append(dict(tag=t, A=z[f"c{i}_A"], B=z[f"c{i}_B"], LB=z[f"c{i}_LB"], RB=z[f"c{i}_RB"],
                          OM=int(z[f"c{i}_OM"]), cols=z[f"c{i}_cols"]))
    return cases


@pytest.fixture(scope="session")
def golden():
    return load_golden()
