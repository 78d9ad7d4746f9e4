import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _build_if_missing():
    """a fresh checkout has no librmcv_hip.so (build products are not in git): build it, as __graft_entry__.build() does"""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "rmcv_amd", "lib", "librmcv_hip.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "rmcv_amd", "csrc"), "-j", str(min(8, os.cpu_count() or 1))], check=True)


def pytest_sessionstart(session):
    _build_if_missing()
    """torch bundles its own HIP runtime; when librmcv_hip.so (system ROCm) initialises the GPU first, a later
    torch.cuda initialisation in the same process reports "No HIP GPUs".  Tests that hand torch tensors to the
    library need both, so torch gets the device first (as bench.py does by construction)."""
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.cuda.init()
    except Exception:
        pass


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
