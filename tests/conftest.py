import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def golden_path(name):
    return os.path.join(ROOT, "tests", "golden", name + ".npz")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(golden_path(name), allow_pickle=False)

    return load
