import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure oracle/libddn_oracle.so and libdsdneo_hip.so exist (build is idempotent and quick)."""
    import __graft_entry__ as g
    g.build()
    return True


def golden(name):
    import numpy as np
    return np.load(os.path.join(HERE, "golden", name))
