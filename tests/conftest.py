import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hierarchical-3d-gaussians_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run under gpurun)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from hgs import _lib
    assert _lib.lib().hgs_device_count() >= 1, "libhgs.so sees no HIP device"
    return torch.device("cuda:0")
