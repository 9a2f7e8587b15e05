import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this container")
    import msod_amd  # noqa: F401
    from msod_amd import _lib
    lib = _lib.load()            # raises if the .so has not been built: GPU tests must not silently skip
    _lib.check(lib.cft_device_check(), "cft_device_check")
    return torch.device("cuda:0")
