import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
# The GPU tests run on the DEFAULT switch set - the binary path bench.py times: no TRICOLO_* variable is set
# here.  Opt-in and forced plans are covered by child-process runs (test_opt_in_kernels_child_process,
# test_halo_kernels_ab_switch in test_gpu_ops.py), each of which sets its switches in the child's environment only.


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
