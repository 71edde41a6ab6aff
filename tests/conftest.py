import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def rq():
    """The product: python host mirror over the C-ABI (fails loudly without the HIP library)."""
    import rayuela_jl_amd
    return rayuela_jl_amd
