import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle dominates the wall time of the parity tests.  On a many-core host (the GPU boxes report 256 hardware threads) torch's default
    # intra-op thread count makes it 2-4 x SLOWER than 16-32 threads do (bench.py's cpu_baseline sweep: 8 / 16 / 32 / 64 threads = 2.3 / 2.9 / 3.0 /
    # 1.6 pairs/s) and the suite's time then depends on the host's other tenants: cap it.  Small hosts (the 8-CPU build container) are unaffected.
    if (os.cpu_count() or 1) > 32:
        import torch
        torch.set_num_threads(int(os.environ.get("CFT_TEST_CPU_THREADS", "32")))


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
