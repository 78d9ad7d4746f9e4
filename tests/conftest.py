import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    oracle_lib.set_math_mode(0)  # pinned math: the parity contract with the GPU
    return oracle_lib


@pytest.fixture(scope="session")
def ctx():
    """a GPU context; the HIP library must load and find a device -- no fallback"""
    from rmcv_amd import Context
    c = Context(device=0, max_frames=32)
    yield c
    c.close()
