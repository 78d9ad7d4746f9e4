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
    ap.add_argument("--steps", type=int, default=1)
    a = ap.parse_args()
    import bench
    world, rank, _, launched = bench.resolve_world(a, os.environ)      # the check bench.py itself applies
    assert launched
    dist.init_process_group("gloo")
    import oracle_lib as O
    from rmcv_amd import dist as rdist
    from rmcv_amd import synth
    O.set_math_mode(0)
    ap_steps = a.steps
    # the bench's pipelined form: `ap_steps` steps through a ring of `depth` records, every record gathered by the hook the bench
    # hands to rmcv_pipeline_set_hook (rmcv_amd.dist.TorchGatherHook: torch.distributed.gather with async_op) -- driven here by a
    # stand-in for the ring with the same hook protocol (hook(ticket, record pointer, bytes, stream) right behind the record's
    # production; a record is refilled only after the gather that last read it is through); step s of rank r detects frames
    # [(s*world + r)*n, ...)
    depth = 2
    _, rb = rdist.record_layout(a.frames, a.cap)
    ring = [np.zeros(rb, np.uint8) for _ in range(depth)]            # the records (host memory here, HBM on the GPU box)
    hook = rdist.TorchGatherHook(rb, depth, "cpu")
    results = []

    def consume(ticket):
        hook.wait(ticket)
        if rank == 0:
            results.append(rdist.unpack_records(hook.records(ticket), a.frames, a.cap))   # consumed before the buffers are reused
    for s in range(ap_steps):
        k = s % depth
        if s >= depth:
            consume(s - depth)                                        # (the pipeline orders the rewrite behind the hook's event)
        offs, arms = [0], []
        base = (s * world + rank) * a.frames
        for i in range(base, base + a.frames):                          # bench.py: rank r owns frames [r*n, (r+1)*n) of its step
            x = O.detect_frame(synth.frame(i, 640, 512))["armours"]
            arms.append(x)
            offs.append(offs[-1] + len(x))
        arm = (np.concatenate(arms) if arms else np.zeros(0, O.ARMOUR)).view(np.uint8).reshape(-1)
        ring[k][:] = 0
        rdist.fill_record(rdist.tensor_at(ring[k].ctypes.data, rb, "cpu"), a.frames, a.cap, np.asarray(offs, np.int32), arm)
        assert hook(s, ring[k].ctypes.data, rb, 0) is None
    for s in range(max(0, ap_steps - depth), ap_steps):               # drain, in step order
        consume(s)
    if rank == 0:
        g, goffs = results[0]
        h = hashlib.sha256()
        for gg, _ in results:
            h.update(gg.tobytes())
        print(json.dumps({"n_gpus": dist.get_world_size(), "armours_gathered": int(g.shape[0]), "frame_offs": goffs.tolist(),
                          "sha256": hashlib.sha256(g.tobytes()).hexdigest(), "steps": len(results),
                          "armours_all_steps": int(sum(gg.shape[0] for gg, _ in results)), "sha256_all_steps": h.hexdigest()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
