import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Every GPU test runs with guard bands behind the library's scratch tensors, forward / backward buffers, generator
# state and packed weights (movenet_amd.ops._GuardBands, RingGenerator.check_errors): a kernel that writes past the
# end of one fails the test that launched it instead of corrupting whichever tensor the allocator placed next
# (round 3 found such a write that way: DESIGN.md section 4.4).
os.environ.setdefault("MOVENET_DEBUG_GUARD", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load
