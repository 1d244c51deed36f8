import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the library reads no environment variable: the tests' KOMB_* switches reach it as per-context options through the
    # Python plumbing (komb_amd/api.py: FORWARD_ENV_OPTIONS)
    import komb_amd.api
    komb_amd.api.FORWARD_ENV_OPTIONS = True


@pytest.fixture(scope="session")
def built():
    """Build libkomb_accel.so and the oracle once per session."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def golden(built):
    import json
    path = os.path.join(ROOT, "tests", "golden", "graphs.json")
    with open(path) as f:
        return json.load(f)
