import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_small():
    return dict(np.load(os.path.join(GOLDEN_DIR, "golden_small.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden_mid():
    return dict(np.load(os.path.join(GOLDEN_DIR, "golden_mid.npz"), allow_pickle=False))
