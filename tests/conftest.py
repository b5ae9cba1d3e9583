import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(ROOT / "tests" / "golden" / f"{name}.npz", allow_pickle=False)

    return load


@pytest.fixture(scope="session")
def _poison_tracker():
    """Guard bands around and NaN fill inside every buffer the product (and the kernel-level tests) allocate on the GPU (tests/poison.py):
    on wherever a GPU is present (the whole `-m gpu` suite runs under it), FK_TEST_POISON=0 switches it off."""
    import torch
    if os.environ.get("FK_TEST_POISON", "1") == "0" or not torch.cuda.is_available():
        return None
    from tests import poison
    return poison.install()


@pytest.fixture(autouse=True)
def _poison_check(request, _poison_tracker):
    yield
    if _poison_tracker is not None:
        bad = _poison_tracker.check()
        assert not bad, f"a kernel wrote outside its buffer (allocation index, bytes, side): {bad}"
