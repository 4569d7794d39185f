import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def gpu():
    """Initialises libdbhip on device 0; fails (never skips) when the HIP path is unavailable."""
    from databend_amd import device
    device.init(0)
    return device
