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


def far_tol(base, amp=1.0):
    """Tolerance of a comparison in which the far field's truncation takes part (far-field mode against the exact mode,
    a spectral shard -- whose boxes start at its own first point -- against the whole grid): `base` (the bound the test
    held at expansion degree 22: rounding) or amp x the library's truncation bound 18 theta^-(degree + 1), whichever is
    larger (sr_far_field_truncation_bound: 1.6e-11 at the default degree 19; 2.6e-13 with -DSR_KFD=22, where `base`
    decides).  amp > 1: measures relative to a NET quantity whose parts cancel (the bound is relative to a line's own
    contribution)."""
    from spectrobot_amd import engine
    return max(float(base), float(amp) * engine.far_field_truncation_bound())


def relerr(a, b):
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    den = np.maximum(np.abs(b), np.finfo(float).tiny)
    return float(np.max(np.abs(a - b) / den)) if a.size else 0.0
