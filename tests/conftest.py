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
    # The oracle issues thousands of small per-tile tensor ops; on a 256-thread host torch's default thread pool turns
    # each of them into a fork-join over every core (the 40-step PSNR case took 8 minutes that way, 1 minute with 8
    # threads).
    import torch
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16)))


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from hgs import _lib
    assert _lib.lib().hgs_device_count() >= 1, "libhgs.so sees no HIP device"
    return torch.device("cuda:0")
