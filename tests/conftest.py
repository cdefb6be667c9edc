import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import ma_amd
    n = ma_amd.device_count()
    if n < 1:
        pytest.fail("no HIP device visible: the -m gpu tests must run on the GPU box")
    ma_amd.set_device(0)
    return 0
