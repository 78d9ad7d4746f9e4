"""Every code path of the kernels against the oracle, including the ones the default configuration rarely takes:
the literal contour scanner and the mid tier forced on every frame, and another grid size of k_binary."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("env", [{}, {"RMCV_CONTOURS_LITERAL": "1"}, {"RMCV_CONTOURS_LITERAL": "2"}, {"RMCV_K1_BPC": "2"}])
def test_variant(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_variant_check.py")], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "variant ok" in r.stdout
    if env.get("RMCV_CONTOURS_LITERAL") == "1":
        assert "slow-path frames: 4" in r.stdout
    if env.get("RMCV_CONTOURS_LITERAL") == "2":
        assert "slow-path frames: 0 mid-tier frames: 4" in r.stdout
