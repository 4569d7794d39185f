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
    # Some GPU tests drive plans through torch.distributed, and PyTorch ships its own librccl.so; libdbhip's communicator (dlopen)
    # adopts an RCCL the process already holds (k_comm.hip load_rccl). Loading torch FIRST makes every test process end up with exactly
    # one RCCL whatever the order of the test files — two copies in one process corrupt the heap at exit.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    from databend_amd import device
    device.init(0)
    return device
