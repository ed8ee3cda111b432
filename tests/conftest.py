import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def refvec():
    """Known answers held by the reference's own tests/examples (tests/golden/reference_vectors.json)."""
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def npvec():
    """numpy/scipy full-precision vectors (tests/golden/make_golden.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "numpy_scipy_vectors.npz"))


@pytest.fixture(scope="session")
def blvec():
    """Independent truths at the BASELINE lane lengths 4096 / 8192 / 16384 / 512 (tests/golden/make_golden_baseline.py):
    numpy / scipy on every lane, long-double O(n^2) definitions on lane 0, mpmath on 12 bins of lane 0."""
    return np.load(os.path.join(ROOT, "tests", "golden", "baseline_lengths.npz"))


@pytest.fixture(autouse=True)
def _reload_library_switches():
    """The library parses its NDFFT_* switches once (csrc/switches.h); a test that changed the environment (monkeypatch.setenv + reload_switches)
    must not leak its setting into the next test: reload after every test, for whichever builds of the library this process has loaded."""
    yield
    lib_mod = sys.modules.get("ndrustfft_amd._lib")
    if lib_mod is None:
        return
    for lib in list(getattr(lib_mod, "_loaded", [])):
        try:
            lib.c.ndfft_reload_switches()
        except Exception:
            pass
