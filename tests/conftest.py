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


def pytest_collection_modifyitems(config, items):
    """`gpu` tests (and the C-ABI example, which launches kernels) are skipped, not failed, on a box without a GPU or
    without the built library -- plain `pytest tests` then behaves like `-m "not gpu"`."""
    import torch

    from event_based_bos_amd import _hip

    if torch.cuda.is_available() and os.path.exists(os.environ.get("EBOS_HIP_LIBRARY", _hip.LIB_PATH)):
        return
    skip = pytest.mark.skip(reason="needs an MI355X and the built libebos_hip.so")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _default_policies(monkeypatch):
    """The tests pin the DEFAULT behaviour (resident solver loop, deferred idiom results, ...): policy variables a shell may carry
    are taken out of every test's environment; a test that wants one sets it itself."""
    for var in ("EBOS_RESIDENT", "EBOS_RESIDENT_MAX_IMBALANCE", "EBOS_FUSE_API", "EBOS_STRICT"):
        monkeypatch.delenv(var, raising=False)


@pytest.fixture(scope="session")
def golden_small():
    return dict(np.load(os.path.join(GOLDEN_DIR, "golden_small.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden_mid():
    return dict(np.load(os.path.join(GOLDEN_DIR, "golden_mid.npz"), allow_pickle=False))
