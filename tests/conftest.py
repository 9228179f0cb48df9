import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (vq_voice_swap_amd/__init__.py), before the HIP runtime starts

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


@pytest.fixture(scope="session")
def lib_built():
    """The HIP library must exist (it is built by __graft_entry__.build(); hipcc cross-compiles on CPU)."""
    from vq_voice_swap_amd import _native

    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    return _native.lib()
