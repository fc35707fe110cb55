import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))

    return load


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """A checkout without the in-tree libfcl_hip.so (it is git-ignored): compile it once per session before any test needs it."""
    if not os.path.exists(os.path.join(ROOT, "fcl-taco2_amd", "libfcl_hip.so")):
        import __graft_entry__ as ge

        ge.build()
