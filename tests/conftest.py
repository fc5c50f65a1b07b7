import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
# conv_wgrad_c64_kernel is planned from 6 row tiles per CU on (conv_wgrad.hip c64_wgrad_geometry; 12 for most of round 3); the exactness cases
# c64_32 / c64_56 of test_gpu_ops.py are sized for 6: pinned here so that they keep covering the kernel if the default moves again
os.environ.setdefault("TRICOLO_C64_MIN_TILES_PER_CU", "6")
os.environ.setdefault("TRICOLO_S2F_CONV", "1")            # conv_s2f_kernel is opt-in (measured slower in the step): the tests keep it covered


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
        return cache[name]
    return load
