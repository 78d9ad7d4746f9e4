"""The N>1 path on CPU: world_size 2, gloo.  Frames are sharded contiguously, each rank "detects" its shard
(with the CPU oracle standing in for the GPU here), and the one gather of the path must reproduce the
single-process armour list byte for byte."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
N_FRAMES, W, H = 8, 640, 512


def detect_shard(lo, hi):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import oracle_lib as O
    from rmcv_amd import synth
    offs, arms = [0], []
    for i in range(lo, hi):
        a = O.detect_frame(synth.frame(i, W, H))["armours"]
        arms.append(a)
        offs.append(offs[-1] + len(a))
    arm = np.concatenate(arms) if arms else np.zeros(0, O.ARMOUR)
    return np.asarray(offs, np.int32), arm.view(np.uint8).reshape(-1)


def worker(rank, world, port, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from rmcv_amd import dist as rdist
    lo, hi = rdist.shard(N_FRAMES, rank, world)
    offs, arm = detect_shard(lo, hi)
    try:
        res = rdist.gather_detections(offs, arm, cap)
        if rank == 0:
            q.put(("ok", res[0].tobytes(), res[1].tolist()))
    except OverflowError as e:
        if rank == 0:
            q.put(("overflow", str(e), None))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run(world, cap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    ps = [ctx.Process(target=worker, args=(r, world, port, cap, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = q.get(timeout=180)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_gather_two_ranks_equals_single():
    offs, arm = detect_shard(0, N_FRAMES)
    assert offs[-1] > 0
    kind, data, goffs = run(2, cap=64)
    assert kind == "ok"
    assert data == arm.tobytes()
    assert goffs == offs.tolist()


def test_gather_reports_overflow():
    kind, msg, _ = run(2, cap=0)
    assert kind == "overflow" and "capacity" in msg


def test_shard_and_record_layout():
    sys.path.insert(0, ROOT)
    from rmcv_amd import dist as rdist
    assert [rdist.shard(2048, r, 8) for r in (0, 7)] == [(0, 256), (1792, 2048)]
    assert sum(hi - lo for lo, hi in (rdist.shard(10, r, 4) for r in range(4))) == 10
    head, total = rdist.record_layout(256, 4096)
    assert head % 16 == 0 and head >= 257 * 4 and total == head + 4096 * 88
    rec = rdist.new_record(3, 2, "cpu")
    rdist.fill_record(rec, 3, 2, np.array([0, 1, 1, 2], np.int32), np.arange(176, dtype=np.uint8))
    arm, offs = rdist.unpack_records([rec], 3, 2)
    assert arm.shape == (2, 88) and offs.tolist() == [0, 0, 1, 1, 2][:0] + [0, 1, 1, 2]
