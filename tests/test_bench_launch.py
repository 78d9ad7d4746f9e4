"""`python bench.py --gpus N` must itself start N ranks (or fail loudly) -- never run one GPU and call it N.
CPU tests of the launcher: world 2, gloo, the oracle standing in for the per-rank detector."""
import hashlib
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_resolve_world_refuses_a_mismatch():
    a = types.SimpleNamespace(gpus=8)
    assert bench.resolve_world(a, {}) == (1, 0, 0, False)            # plain invocation: main() goes on to launch_ranks
    with pytest.raises(SystemExit) as e:
        bench.resolve_world(a, {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert "WORLD_SIZE=2" in str(e.value)
    with pytest.raises(SystemExit):                                   # and the other way round: launched as 2, asked for 1
        bench.resolve_world(types.SimpleNamespace(gpus=1), {"RANK": "1", "WORLD_SIZE": "2", "LOCAL_RANK": "1"})
    assert bench.resolve_world(types.SimpleNamespace(gpus=2), {"RANK": "1", "WORLD_SIZE": "2", "LOCAL_RANK": "1"}) == (2, 1, 1, True)


def test_plain_gpus_n_without_n_gpus_fails_loudly():
    """this container has no GPU: `bench.py --gpus 2` must exit non-zero and say why (round 1 ran one GPU and printed n_gpus 1)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box: the launch would really happen")
    assert p.returncode != 0
    assert b"--gpus 2 but only" in p.stderr and b"n_gpus" not in p.stdout


def test_launch_ranks_world2_equals_single_process():
    """the launcher starts 2 ranks; the list gathered through the bench's record path equals the concatenation of the
    single-process lists of the two shards, byte for byte (the property C4 needs at 8 x 256 frames)"""
    import oracle_lib as O
    from rmcv_amd import synth
    O.set_math_mode(0)
    frames = 3
    rc, line = bench.launch_ranks(2, ["--gpus", "2", "--frames", str(frames)], script=os.path.join(HERE, "_bench_stub_worker.py"),
                                  timeout=600)
    assert rc == 0 and line
    out = json.loads(line)
    assert out["n_gpus"] == 2
    arms, offs = [], [0]
    for i in range(2 * frames):
        a = O.detect_frame(synth.frame(i, 640, 512))["armours"]
        arms.append(a)
        offs.append(offs[-1] + len(a))
    whole = np.concatenate(arms).view(np.uint8)
    assert out["armours_gathered"] == offs[-1] > 0
    assert out["frame_offs"] == offs
    assert out["sha256"] == hashlib.sha256(whole.tobytes()).hexdigest()


def test_launch_ranks_reports_a_failing_rank():
    rc, line = bench.launch_ranks(2, ["--gpus", "3"], script=os.path.join(HERE, "_bench_stub_worker.py"), timeout=600)
    assert rc != 0 and line is None


def test_pipelined_async_gathers_world2_keep_every_step_apart():
    """the default gather of a multi-rank bench run: torch.distributed.gather with async_op over alternating records, a record
    rewritten only after work.wait() -- five steps, every step's gathered list equals the single-process list of ITS frames"""
    import oracle_lib as O
    from rmcv_amd import synth
    O.set_math_mode(0)
    frames, steps = 2, 5
    rc, line = bench.launch_ranks(2, ["--gpus", "2", "--frames", str(frames), "--steps", str(steps)],
                                  script=os.path.join(HERE, "_bench_stub_worker.py"), timeout=600)
    assert rc == 0 and line
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == steps
    h, total = hashlib.sha256(), 0
    for i in range(steps * 2 * frames):                                # step s covers frames [s*2*frames, (s+1)*2*frames) in rank order
        a = O.detect_frame(synth.frame(i, 640, 512))["armours"]
        h.update(a.view(np.uint8).tobytes())
        total += len(a)
    assert out["armours_all_steps"] == total > 0
    assert out["sha256_all_steps"] == h.hexdigest()


def test_default_schedule_is_the_measured_one():
    """8 batches in flight, 2 pixel + 4 sparse streams, 12 hardware queues (found by alternating schedule shapes
    inside one process); the environment's GPU_MAX_HW_QUEUES wins when it is set"""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    saved = os.environ.pop("GPU_MAX_HW_QUEUES", None)
    try:
        bench = importlib.reload(importlib.import_module("bench"))
        a = bench.parse_args([])
        assert (a.streams, a.pixel_streams, a.sparse_streams, a.gpus) == (8, 2, 4, 1)
        assert os.environ["GPU_MAX_HW_QUEUES"] == "12"
        assert bench.parse_args(["--workload", "c5"]).streams == 8
    finally:
        if saved is not None:
            os.environ["GPU_MAX_HW_QUEUES"] = saved


def test_two_builds_of_the_library_can_be_loaded_side_by_side():
    """abi.load / abi.use (bench.py RMCV_BENCH_AB=lib:...): a second handle does not replace THE library until asked to"""
    from rmcv_amd import abi
    first = abi.lib()
    other = abi.load(abi.LIB_PATH)
    assert abi.lib() is first and other.rmcv_abi_version() == first.rmcv_abi_version()
    prev = abi.use(other)
    try:
        assert prev is first and abi.lib() is other
    finally:
        abi.use(first)
    assert abi.lib() is first
