import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# MIOpen's default hybrid find may settle on a different convolution solver from one call to the next while it refines
# its choice; the bit-for-bit comparisons (eager vs HIP-graph replay, repeatability) need one solver per shape, as
# bench.py pins it.  Must be set before the first convolution of the process.
os.environ.setdefault("MIOPEN_FIND_MODE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    return load_golden
