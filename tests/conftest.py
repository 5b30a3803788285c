import lzma
import os
import sys

import numpy as np
import pytest

try:
    # Load order matters when torch shares the process (two tests keep their input in torch tensors):
    # the PyTorch wheel bundles its own ROCm runtime under the same SONAMEs as /opt/rocm's, and whichever
    # libamdhip64 is mapped first serves both.  torch cannot start on /opt/rocm's copy, libpwpp_hip.so is
    # happy with either, so torch goes first.
    import torch  # noqa: F401
except ImportError:
    pass

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "patchwork-plusplus_amd", "python"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_kitti_cache = {}


def load_kitti(k):
    """KITTI sample frame k (0..5), byte-identical to /root/reference/data/%06d.bin."""
    if k not in _kitti_cache:
        with lzma.open(os.path.join(HERE, "golden", "kitti_%06d.bin.xz" % k), "rb") as f:
            _kitti_cache[k] = np.frombuffer(f.read(), np.float32).reshape(-1, 4).copy()
    return _kitti_cache[k]


@pytest.fixture(scope="session")
def kitti():
    return [load_kitti(k) for k in range(6)]


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(HERE, "golden", "kitti_golden.npz"))


@pytest.fixture(scope="session")
def oracle_built():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


def ground_mask(idx, n):
    m = np.zeros(n, np.uint8)
    m[idx] = 1
    return m
