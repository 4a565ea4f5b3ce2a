import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def po():
    """The CPU oracle (test infrastructure)."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def meshes(po):
    return po.default_meshes()


@pytest.fixture(scope="session")
def golden_frames():
    return dict(np.load(os.path.join(GOLDEN, "frames_240x135.npz")))


@pytest.fixture(scope="session")
def golden_bvh():
    return dict(np.load(os.path.join(GOLDEN, "bvh.npz")))


@pytest.fixture(scope="session")
def golden_kat():
    return dict(np.load(os.path.join(GOLDEN, "kat.npz")))


@pytest.fixture(scope="session")
def golden_hashes():
    return json.load(open(os.path.join(GOLDEN, "hashes.json")))


def bits(a):
    """Bit pattern view for exact float comparison (distinguishes -0.0, compares NaNs by payload)."""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
