"""BASELINE config 4's exchange at WORLD SIZE 2 on the one GPU this build has: two processes, each with a pipeline whose built-in
gather (rmcv_pipeline_set_gather -> rmcv_gather) runs on a two-rank rmcv_comm.  RCCL cannot put two ranks on one device, so
librccl.so.1 is the stand-in of tests/fake_rccl (payloads through files, pairing in EXECUTION order like RCCL's): what is under test
is the library's side -- the root / peer roles of rmcv_gather, the placement of the records by rank, and the pipeline's event chain
that keeps one communicator's gathers in ticket order although consecutive tickets finish on different streams."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from rmcv_amd import CAMP_BLUE, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_gather_every_ticket_in_order(oracle, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    libdir = tmp_path / "fake_rccl"
    libdir.mkdir()
    subprocess.run([hipcc, "-O1", "-shared", "-fPIC", "-o", str(libdir / "librccl.so.1"), os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")],
                   check=True, timeout=300)
    meet = tmp_path / "meet"
    meet.mkdir()
    nb, world, n, w, h, depth = 9, 2, 12, 640, 512, 3
    env = dict(os.environ, LD_LIBRARY_PATH=str(libdir) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), FAKE_RCCL_DIR=str(meet))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gather_world2_worker.py"), str(r), str(world), str(meet), str(nb)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, outs[r][-3000:])
    res = [np.load(meet / ("rank%d.npz" % r)) for r in range(world)]
    rb, ao = int(res[0]["record_bytes"]), int(res[0]["armours_offset"])
    g = res[0]["gathered"]                                         # [ticket][world x record bytes] as the ROOT received them
    assert g.shape == (nb, world * rb)
    for r in range(world):
        counts, offs = res[r]["counts"], res[r]["offs"]
        arms = res[r]["armours"]
        pos = 0
        for j in range(nb):
            rec = g[j, r * rb:(r + 1) * rb]
            k = int(counts[j])
            mine = arms[pos:pos + 88 * k]
            pos += 88 * k
            # rank r's record of ticket j, as the root holds it == what rank r itself collected for ticket j ...
            assert rec[:4 * (n + 1)].view(np.int32).tolist() == offs[j].tolist(), (r, j)
            assert rec[ao:ao + 88 * k].tobytes() == mine.tobytes(), (r, j)
            # ... == the oracle's lists for rank r's frames of that batch
            fr = synth.batch(500000 + 10007 * r + 131 * j, n, w, h, CAMP_BLUE, j % 2, threads=4)
            ref = np.concatenate([oracle.detect_frame(f)["armours"] for f in fr])
            assert mine.tobytes() == ref.tobytes(), (r, j)
