import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The parity oracle (oracle/liboracle.so, built on demand with g++)."""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from webdgs_amd import ops
    dev = ops.HipDevice(0)
    yield dev
    dev.synchronize()
