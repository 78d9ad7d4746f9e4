"""Stand-in for bench.py's per-rank body, started by bench.launch_ranks in tests/test_bench_launch.py (CPU, gloo).

Every rank "detects" its contiguous shard of the frame stream (the CPU oracle stands in for the GPU), packs the lists into the
record rmcv_batch_compact_armours writes on the device, and the records travel through the bench's own gather path
(rmcv_amd.dist.gather_records -> unpack_records).  Rank 0 prints ONE JSON line, as bench.py does."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, required=True)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--cap", type=int, default=64)
    a = ap.parse_args()
    import bench
    world, rank, _, launched = bench.resolve_world(a, os.environ)      # the check bench.py itself applies
    assert launched
    dist.init_process_group("gloo")
    import oracle_lib as O
    from rmcv_amd import dist as rdist
    from rmcv_amd import synth
    O.set_math_mode(0)
    offs, arms = [0], []
    for i in range(rank * a.frames, (rank + 1) * a.frames):             # bench.py: rank r owns frames [r*n, (r+1)*n)
        x = O.detect_frame(synth.frame(i, 640, 512))["armours"]
        arms.append(x)
        offs.append(offs[-1] + len(x))
    arm = (np.concatenate(arms) if arms else np.zeros(0, O.ARMOUR)).view(np.uint8).reshape(-1)
    rec = rdist.fill_record(rdist.new_record(a.frames, a.cap, "cpu"), a.frames, a.cap, np.asarray(offs, np.int32), arm)
    recs = rdist.gather_records(rec, out=rdist.new_gather_list(rec))
    if rank == 0:
        g, goffs = rdist.unpack_records(recs, a.frames, a.cap)
        print(json.dumps({"n_gpus": dist.get_world_size(), "armours_gathered": int(g.shape[0]), "frame_offs": goffs.tolist(),
                          "sha256": hashlib.sha256(g.tobytes()).hexdigest()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
