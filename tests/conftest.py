import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_index():
    from tests.util import load_golden_index
    return load_golden_index()


@pytest.fixture(scope="session")
def golden_reads():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "tiny_reads.npz"))


@pytest.fixture(scope="session")
def golden_primitives():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "primitives.npz"), allow_pickle=True)
