import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: needs a real MI355X and minutes of host time (opt in with `-m gpu_slow`; not part of `-m gpu`)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        return cache[name]

    return load


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (C restatement, built with gcc on first use)"""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def hip_lib():
    """libbgflow_amd.so -- built in-tree by __graft_entry__.build(); build it here if missing"""
    from bgflow_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_extension()
    return _lib.lib()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
