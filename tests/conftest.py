import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name is not a Python identifier)."""
    return importlib.import_module("nano-vllm-go_amd")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement of the reference (test infrastructure only)."""
    from oracle import purego_oracle
    purego_oracle.lib()
    return purego_oracle


@pytest.fixture(scope="session")
def gpu(pkg):
    """Loads the HIP library and insists on a device: gpu tests must never pass on a fallback."""
    L = pkg.lib()
    n = L.nvl_device_count()
    assert n >= 1, "no HIP device visible: -m gpu tests need the MI355X box"
    return pkg


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
