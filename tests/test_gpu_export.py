"""The per-frame chain returns its lists through pinned staging buffers.  By default the last kernel of the chain stores them there
itself (ExportArgs / k_export, rmcv_host.hip); RMCV_EXPORT=0 brings back the row of small device-to-host copies of round 2.  Both
must hand the caller the same bytes (plain, stress and a dense frame whose lists exceed the copy windows)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_export_by_kernel_and_by_copies_return_the_same_bytes():
    outs = []
    for mode in ("1", "0"):
        cp = subprocess.run([sys.executable, os.path.join(HERE, "_chain_worker.py")], env=dict(os.environ, RMCV_EXPORT=mode),
                            capture_output=True, text=True, timeout=300)
        assert cp.returncode == 0, cp.stderr[-2000:]
        outs.append([ln for ln in cp.stdout.splitlines() if ln.startswith("chain ")][-1])
    assert outs[0] == outs[1]
