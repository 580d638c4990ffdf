import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")
    # a fresh checkout has no libspectrobot_hip.so (built artefacts are not in history) and an edited
    # source tree has a stale one: (re)build it when it is missing or older than its sources (hipcc
    # cross-compiles gfx950 without a GPU).  Loaded by path -- the package refuses to import without
    # a library that exports every symbol of the header.
    import importlib.util
    spec = importlib.util.spec_from_file_location("_sr_build", os.path.join(ROOT, "spectrobot_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


def relerr(a, b):
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    den = np.maximum(np.abs(b), np.finfo(float).tiny)
    return float(np.max(np.abs(a - b) / den)) if a.size else 0.0
