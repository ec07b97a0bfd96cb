import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # TEST INFRASTRUCTURE: SNAPGPU_TEST_LIB=<path> makes the Python mirror used BY THE TESTS open another build of the C ABI -- the
    # wavefront emulator's (tests/emu/_build/libsnapgpu_emu.so), to run the `-m gpu` tests on the host, or an A/B build.  The product
    # itself has no such switch: snap_amd.aligner opens snap_amd/libsnapgpu.so only.
    alt = os.environ.get("SNAPGPU_TEST_LIB")
    if alt:
        import snap_amd.aligner as al
        al.LIB_PATH = os.path.abspath(alt)
        al._lib = None


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
