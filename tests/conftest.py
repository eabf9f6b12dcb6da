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
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def unpack_golden_active(rec, m):
    return np.unpackbits(rec["active"], axis=1)[:, :m].astype(bool)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
