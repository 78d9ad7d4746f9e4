#!/usr/bin/env python3
"""bench.py -- frames/s of the armour-detection hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" = one pass of the hot path over one batch of 256 synthetic 1280x1024 BGR frames that are
already resident in HBM: rm::extract_color -> rm::filter_lightblobs -> rm::filter_armours
(reference executable/main.cpp:172-176), followed by the device-side compaction of the armour
lists and -- for N > 1 -- the RCCL gather of those lists to rank 0 (BASELINE config 4).  Weak
scaling: every rank owns its own 256 frames, no collective on the data path.

Eight batches are in flight (own context, own frames each), pixel kernels alternating over 2 streams, the sparse stages over 4.

Prints ONE JSON line on rank 0: the contract fields plus
  roofline      k_binary (the kernel that moves the algorithmic 4 B/px), timed with HIP events on
                its own launch stream inside this process, COLD: every launch on another context's
                frames and buffers (the same gigabyte again would come partly out of the 256 MB
                Infinity Cache -- printed beside it as same_frames_every_launch)
  cpu_baseline  the CPU oracle (a port/restatement of the reference path, oracle/) timed on this
                box's host cores on a bounded sample of the same frames (rank 0, N=1 only)

Dev tool: RMCV_BENCH_AB="<option>:<a>:<b>" | "sched:<ctx,pix,sparse>:<...>" | "lib:<another build>" alternates two settings between
regions of ONE process (two processes of the same command differ by +-3 % on one box; regions inside a process by 0.1 %).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A step's kernels must not share a
# queue with another batch's or with RCCL's stream: the null stream + three batch streams + RCCL's own need five, and with four
# the gather of every step cost 15 % (736 k against 843 k frames/s, tools/ab_dist.sh).  Read by the HIP runtime at start-up.
# The steps run 8 batches in flight over 2 pixel + 4 sparse streams (below): with the null stream and RCCL's that is more than the
# default of 4 hardware queues and than round 2's 6 -- 12 (8 and 16 measure the same).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

FRAMES = 256
WORKLOADS = {"c3": (1280, 1024), "c5": (1920, 1200),   # BASELINE.json configs[2] (the metric's config) and configs[4]
             "legacy": (1280, 1024)}                     # FindLightBlobs (minAreaRect boxes, camp vote) in place of filter_lightblobs
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=7,
                    help="the timed region (exactly --steps steps between two barriers) is repeated this many times; value = median")
    ap.add_argument("--warmup-seconds", type=float, default=0.4,
                    help="besides --warmup steps: untimed steps for at least this long, so the clocks have ramped before the timed region")
    ap.add_argument("--frames", type=int, default=FRAMES, help="frames per GPU per step")
    ap.add_argument("--variant", type=variant_id, default=0,
                    help="synthetic stream: plain (0, the metric's), stress (1), dense1..dense4 (11..14; dense = dense4: +2000 specks and 13 "
                         "bright windows per frame, 5 %% foreground -- frames beyond findContours' LDS tables; a workload beside the metric)")
    ap.add_argument("--handover", action="store_true",
                    help="frame-level hand-over: enqueue every step's sparse kernel beside its own pixel kernel (RMCV_STAGE_HANDOVER)")
    ap.add_argument("--density-sweep", action="store_true",
                    help="after the run: steady-state step time on every density level (plain, dense1..dense4), printed as `density_sweep`")
    ap.add_argument("--cpu-frames", type=int, default=256, help="frames per pass of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the C2 (binary only) side measurements")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3",
                    help="c3: 1280x1024 full path (the metric's config); c5: 1920x1200 full path + SVM digit classify on the icons; "
                         "legacy: c3 with rm::FindLightBlobs(fitEllipse=false) as the blob stage")
    ap.add_argument("--pose", action="store_true",
                    help="add the pose stage (rm::solve_PnP + world position per armour, SURVEY 8f-3) to every step")
    ap.add_argument("--streams", type=int, default=8,
                    help="contexts (buffer sets = batches in flight) the steps are pipelined over (1 = strictly serial steps).  8 over 4 sparse "
                         "streams since the end of round 3: alternating regions of ONE process (RMCV_BENCH_AB=sched:...) put it 4.3-4.5 %% ahead "
                         "of round 2's 4 over 2 on C3 and 1 %% on C5; whole processes, five alternations: 0.2587 against 0.2689 ms per step")
    ap.add_argument("--mode", choices=("pipeline", "alternate"), default="pipeline",
                    help="pipeline (default): the pixel kernels of consecutive steps alternate over --pixel-streams streams, the sparse "
                         "stages run on --sparse-streams higher-priority streams, a step's two halves chained by events; alternate: "
                         "whole steps on one stream per context (round 1's schedule: the same steady state, a longer ramp)")
    ap.add_argument("--pixel-streams", type=int, default=2)
    ap.add_argument("--sparse-streams", type=int, default=4)
    ap.add_argument("--gather", choices=("auto", "torch", "abi"), default="auto",
                    help="the armour-list gather of a launched run: torch.distributed.gather (asynchronous; the default for more than "
                         "one rank: rmcv_gather's multi-rank path has not run on hardware yet -- no multi-GPU box was available to this "
                         "build) or rmcv_gather, the C-ABI entry point that calls RCCL itself (what a C++ host uses; the default for a "
                         "launched single rank, where it moves nothing)")
    return ap.parse_args(argv)


VARIANTS = {"plain": 0, "stress": 1, "dense1": 11, "dense2": 12, "dense3": 13, "dense4": 14, "dense": 14}


def variant_id(v):
    return VARIANTS[v] if v in VARIANTS else int(v)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, script=None, timeout=None, extra_env=None):
    """`python bench.py --gpus N` run directly (no WORLD_SIZE in the environment): start the N ranks as fresh child processes --
    the command the driver itself uses for N > 1 -- and relay rank 0's JSON line.  The parent never touches the GPU (no HIP
    call, no librmcv_hip load), so the children are ordinary first users of their devices.  Returns (exit code, last stdout
    line that parses as JSON or None)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this host driver
    env.update(extra_env or {})
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=timeout)
    line = None
    for ln in p.stdout.decode(errors="replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                json.loads(ln)
                line = ln
            except ValueError:
                pass
    return p.returncode, line


def resolve_world(args, environ):
    """(world, rank, local_rank, launched) from the environment torch.distributed.run sets; --gpus must agree with it.
    Raises SystemExit when they differ: `--gpus N` never runs on another number of GPUs than it reports."""
    launched = "RANK" in environ and "WORLD_SIZE" in environ
    world = int(environ.get("WORLD_SIZE", "1")) if launched else 1
    if launched and args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    return world, int(environ.get("RANK", "0")) if launched else 0, int(environ.get("LOCAL_RANK", "0")) if launched else 0, launched


def main():
    args = parse_args()
    world, rank, local_rank, launched = resolve_world(args, os.environ)
    if args.gpus > 1 and not launched:
        import torch                                             # (may call hipGetDeviceCount; harmless: the ranks are fresh children)
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, have))
        rc, line = launch_ranks(args.gpus, sys.argv[1:])
        if line:
            print(line, flush=True)
        if rc == 0 and (not line or json.loads(line).get("n_gpus") != args.gpus):
            raise SystemExit("bench.py: the %d ranks did not report n_gpus=%d" % (args.gpus, args.gpus))
        raise SystemExit(rc)

    W, H = WORKLOADS[args.workload]
    BYTES_PER_FRAME = 4 * W * H      # 3 B/px BGR read + 1 B/px binary written
    import torch
    import torch.distributed as dist

    from rmcv_amd import (CAMP_BLUE, CAMP_RED, MORPH_CLOSE, MORPH_DILATE, STAGE_ALL, STAGE_ARMOURS, STAGE_BINARY, STAGE_BLOBS,
                          STAGE_HANDOVER, STAGE_IDENTITY, STAGE_NO_IMAGE, STAGE_POSE, OPT_HANDOVER, OPT_PIXEL_GROUPS, OPT_SPARSE_WAVES, Context, LegacyParams, default_params, synth)
    from rmcv_amd import dist as rdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the detection path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = launched                                           # started by torch.distributed.run (by the driver or by launch_ranks)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    n = args.frames
    ns = max(1, args.streams)
    nthreads = min(16, os.cpu_count() or 1)
    host = synth.batch(rank * n, n, W, H, CAMP_BLUE, args.variant, threads=nthreads)
    # every batch in flight has its OWN frames (batch k of rank r starts at stream index r*n + k*1000003): steps that overlap
    # in time must not share input, or the later one would be served from the 256 MB Infinity Cache instead of HBM
    frames_k = [torch.from_numpy(host).to(dev)]                  # resident in HBM before any timing
    # At least FOUR frame sets even when fewer batches are in flight: a launch that reads the gigabyte its predecessor has just read
    # gets part of it from the 256 MB Infinity Cache -- k_binary alone, back to back: 0.215 ms on the same frames, 0.240 rotating over
    # two sets, 0.2575 over four (tools/k1_pipe.py) -- which a camera feed never does.  Everything this file measures "alone" (the
    # roofline kernel, the lone batch, the stage split, serial steps) therefore takes the contexts -- one per frame set, each with its own output buffers -- in turn (nxt below).
    n_sets = max(ns, 4)
    for k in range(1, n_sets):
        frames_k.append(torch.from_numpy(synth.batch(rank * n + k * 1000003, n, W, H, CAMP_BLUE, args.variant, threads=nthreads)).to(dev))
    frames = frames_k[0]
    # Steps are double-buffered over `--streams` contexts (own work buffers, own HIP stream, same resident
    # frames): while the sparse stages of step i (contours, fits, pairing: latency-bound, a few waves per CU) run,
    # the HBM-bound pixel kernel of step i+1 streams -- what a continuous camera feed would do.
    # (the dense streams have up to ~2100 contours per frame: beyond the default limit of 2048)
    ctxs = [Context(device=local_rank, max_frames=n, max_width=W, max_height=H, max_contours=(4096 if args.variant >= 10 or args.density_sweep else 2048))
            for _ in range(n_sets)]                                # the first ns carry the steps; all of them the "alone" measurements (nxt below)
    stages = STAGE_ALL | (STAGE_IDENTITY if args.workload == "c5" else 0) | (STAGE_POSE if args.pose else 0)
    if os.environ.get("RMCV_BENCH_STAGES"):                      # dev knob (tools/ab_streams.sh): a partial path is NOT the metric
        stages = int(os.environ["RMCV_BENCH_STAGES"])
    svm = synth.svm_weights() if args.workload == "c5" else None   # svm.xml is not in the reference: seeded stand-in weights
    for k, c in enumerate(ctxs):
        # several batches in flight: 4 wavefronts per frame in the sparse kernel (throughput); a lone batch: 8 (latency)
        c.set_option(OPT_SPARSE_WAVES, int(os.environ.get("RMCV_SPARSE_WAVES", "4" if ns >= 3 else "8")))
        c.set_option(OPT_PIXEL_GROUPS, int(os.environ.get("RMCV_PIXEL_GROUPS", "2" if ns >= 2 else "3")))   # likewise: 2 pixel workgroups per CU when batches overlap, 3 alone
        c.bind_device_frames(frames_k[k].data_ptr(), n, H, W, keepalive=frames_k[k])
        if svm:
            c.svm_load(*svm)
        if args.pose:
            c.pnp_load()                                          # camera constants of executable/main.cpp:7-19
    ctx = ctxs[0]
    rot = [0]

    def nxt():
        """the next context in turn: its frames were last read, and its buffers last written, n_sets launches ago"""
        rot[0] = (rot[0] + 1) % n_sets
        return ctxs[rot[0]]
    params = default_params()                                     # main.cpp:172-176: BLUE, lb 80, close, ...
    legacy = LegacyParams(1.5, 80, 70, 10, 99999, int(os.environ.get("RMCV_LEGACY_FIT", "0"))) if args.workload == "legacy" else None

    def run_path(c, st, hs):
        if legacy is not None:
            c.run_legacy(legacy, params, st, hs)
        else:
            c.run(params, st, hs)
    cap = n * 8                                                   # armours per rank in the gather record (the synthetic stream has ~3 per frame)
    head, _ = rdist.record_layout(n, cap)
    recs_buf = [rdist.new_record(n, cap, dev) for _ in range(ns)]
    gather_out = [rdist.new_gather_list(r) if use_dist else None for r in recs_buf]   # rank 0's receive buffers, one set per stream
    # The gather of a step is enqueued on the stream its record was produced on (a gather stream of its own added stream-to-queue
    # sharing and event hops that cost 0.07 ms per step).  A communicator's operations must execute in ONE order on every rank;
    # the steps' gathers are issued in step order but on alternating streams, so each gather first waits (an event, on the GPU)
    # for the previous step's gather -- ONE communicator: a second RCCL communicator in the process cost 0.07 ms per step
    # by itself (0.335-0.343 against 0.268-0.284 ms, same box).  torch.distributed.gather synchronises the calling stream with the
    # process group's own stream, which in the pipelined schedule holds up the next step's sparse kernels (711-721 k frames/s).
    if args.gather == "auto":
        args.gather = "abi" if world == 1 else "torch"
    abi_gathers = [rdist.AbiGather(local_rank)] if (use_dist and args.gather == "abi") else None
    works = [None] * ns                                          # torch path: the asynchronous gather of the step that last used record k
    ev_gath = [torch.cuda.Event() for _ in range(ns)]
    abi_recv = [abi_gathers[0].new_recv(r) for r in recs_buf] if abi_gathers else None
    gather_note = args.gather if use_dist else None
    if abi_gathers:
        # self-check before anything is timed: the communicator moves a stamped record from every rank to its slot on the root
        probe = torch.full((4096,), rank + 1, dtype=torch.uint8, device=dev)
        rb = abi_gathers[0].new_recv(probe)
        parts = abi_gathers[0].gather(probe, rb, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ok = 1
        if rank == 0 and any(int(pt.min()) != r_ + 1 or int(pt.max()) != r_ + 1 for r_, pt in enumerate(parts)):
            ok = 0
        t_ok = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.broadcast(t_ok, src=0)
        if int(t_ok.item()) != 1:          # never time a path that moved wrong bytes: fall back to torch.distributed.gather, and say so
            abi_gathers[0].close()
            abi_gathers, abi_recv, gather_note = None, None, "torch (rmcv_gather failed its self-check)"

    def gather_step(k, hs, stream_index):
        if not use_dist:
            return [recs_buf[k]]
        if abi_gathers:
            cur = torch.cuda.current_stream()
            if step_no[0] > 1:
                cur.wait_event(ev_gath[(k - 1) % ns])     # the previous step's gather (step_no was advanced already)
            out = abi_gathers[0].gather(recs_buf[k], abi_recv[k], hs)
            ev_gath[k].record(cur)
            return out
        # asynchronous: the process group's stream waits for the record, the calling stream does not wait for the collective
        out, works[k] = rdist.gather_records(recs_buf[k], out=gather_out[k], async_op=True)
        return out
    # one stream per batch in flight (priorities alternate; with GPU_MAX_HW_QUEUES = 6 every stream has its own hardware queue, which is
    # what lets kernels of two steps actually run concurrently
    prios = [int(x) for x in os.environ.get("RMCV_BENCH_PRIOS", "").split(",") if x] or [0, -1]
    streams = [torch.cuda.Stream(device=dev, priority=prios[k % len(prios)]) for k in range(ns)]
    stream, sh, rec = streams[0], streams[0].cuda_stream, recs_buf[0]
    step_no = [0]
    # software pipeline: stream A carries only k_binary (HBM-bound), stream B (higher priority) the sparse stages;
    # step i's sparse chain waits for its own pixel kernel, the pixel kernel of step i+ns waits for the buffers
    sAs = [torch.cuda.Stream(device=dev, priority=0) for _ in range(max(1, args.pixel_streams))]
    sBs = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(max(1, args.sparse_streams))]
    ev_bin = [torch.cuda.Event() for _ in range(ns)]
    ev_done = [torch.cuda.Event() for _ in range(ns)]
    pipelined = args.mode == "pipeline" and ns > 1
    # frame-level hand-over (the sparse kernel beside its own pixel kernel): built, tested, measured equal on this schedule
    # (tools/ab_vs_round2.sh: 0.2493 against 0.2502 ms per step, three alternating runs each) -- off unless asked for
    handover = args.handover or os.environ.get("RMCV_BENCH_HANDOVER", "0") == "1"
    ho = [handover]                                              # (a list: RMCV_BENCH_AB toggles it between regions)
    for c in ctxs:
        c.set_option(OPT_HANDOVER, 1 if handover else 0)

    cur_stages = [stages]

    shape = [ns, len(sAs), len(sBs)]                             # batches in flight, pixel streams, sparse streams IN USE (RMCV_BENCH_AB "sched" narrows them)
    used = [False] * ns

    def step():
        k = step_no[0] % shape[0]
        first_use = not used[k]
        used[k] = True
        step_no[0] += 1
        if not pipelined:
            cx = nxt() if ns < n_sets else ctxs[k]                 # fewer batches in flight than frame sets: the contexts take turns (see n_sets)
            with torch.cuda.stream(streams[k]):
                run_path(cx, cur_stages[0], streams[k].cuda_stream)
                if works[k] is not None:
                    works[k].wait()                        # the record is rewritten: its previous gather must be through (stream-side wait)
                cx.compact_armours_into(recs_buf[k].data_ptr() + head, cap, recs_buf[k].data_ptr(), streams[k].cuda_stream)
                return gather_step(k, streams[k].cuda_stream, k)
        sA = sAs[(step_no[0] - 1) % shape[1]]
        with torch.cuda.stream(sA):
            if not first_use:
                sA.wait_event(ev_done[k])
            ctxs[k].run(params, cur_stages[0] & (STAGE_BINARY | STAGE_NO_IMAGE), sA.cuda_stream)
            ev_bin[k].record(sA)
        sB = sBs[k % shape[2]]
        with torch.cuda.stream(sB):
            if ho[0]:
                # frame-level hand-over: the sparse kernel is enqueued beside its own pixel kernel and takes each frame when its last
                # strip is written (RMCV_STAGE_HANDOVER: the library orders it after what preceded that pixel kernel, not after it)
                run_path(ctxs[k], (cur_stages[0] & ~(STAGE_BINARY | STAGE_NO_IMAGE)) | STAGE_HANDOVER, sB.cuda_stream)
            else:
                sB.wait_event(ev_bin[k])
                if cur_stages[0] & ~(STAGE_BINARY | STAGE_NO_IMAGE):      # (RMCV_BENCH_STAGES=1, a dev knob: pixel kernels only)
                    run_path(ctxs[k], cur_stages[0] & ~(STAGE_BINARY | STAGE_NO_IMAGE), sB.cuda_stream)
            if abi_gathers and not first_use:
                sB.wait_event(ev_gath[k])                  # the record is rewritten: its previous gather (ns steps back) must be through
            if works[k] is not None:
                works[k].wait()                            # likewise on the torch path (a stream-side wait on the collective, ns steps old)
            if not os.environ.get("RMCV_BENCH_NO_COMPACT"):       # (dev knob: what the compaction kernel costs the chain; not the metric)
                ctxs[k].compact_armours_into(recs_buf[k].data_ptr() + head, cap, recs_buf[k].data_ptr(), sB.cuda_stream)
            # the context's buffers are free once its list is compacted: the gather only reads the record, so the pixel kernel
            # that reuses this context does not wait for the collective (it would lengthen the chain the step rate hangs on)
            ev_done[k].record(sB)
            return gather_step(k, sB.cuda_stream, k % len(sBs))

    def barrier():
        # every stream that carried a gather is drained BEFORE the process group's own collectives (barrier, all_reduce) are
        # enqueued: rmcv_gather's communicator and torch's never have kernels resident together -- with streams sharing
        # hardware queues, a recv queued ahead of an all-reduce on one rank and the reverse on another could otherwise wait on each other
        for wk in works:
            if wk is not None:
                wk.wait()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    if os.environ.get("RMCV_BENCH_STAGES"):                      # dev knob: later stages need the planes of a full pass
        for k in range(ns):
            ctxs[k].run(params, STAGE_ALL, streams[k].cuda_stream)
        torch.cuda.synchronize()
    def agree_max(x):
        """the same number on every rank (MAX): ranks must take the same decisions, a step contains a collective"""
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        recs = step()
    barrier()
    # warm-up by time as well: a GPU that idled while the frames were generated has to ramp its clocks; 5 steps are 1.5 ms
    # (round 1: the driver's 20-step run read 822 k frames/s where 100-step runs read 905-950 k)
    tw, warm_steps = time.perf_counter(), 0
    while agree_max(time.perf_counter() - tw) < args.warmup_seconds:
        for _ in range(max(1, args.steps)):
            recs = step()
        warm_steps += max(1, args.steps)
        barrier()
    # the timed region: EXACTLY --steps steps between two (barrier + synchronize), MAX over ranks; repeated --repeats times,
    # value = the median repeat (median and min over the passes)
    rep_dt, enq_dt = [], []
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            recs = step()
        enq_dt.append(time.perf_counter() - t0)       # host time to enqueue the region's steps (the GPU may still be running them)
        barrier()
        rep_dt.append(agree_max(time.perf_counter() - t0))
    srt = sorted(rep_dt)
    dt = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    ms_per_step = dt / args.steps * 1e3
    value = world * n * args.steps / dt
    # beside the metric: ONE long region (the same loop, 25 x --steps steps).  A timed region starts with an empty pipeline and ends by
    # draining it -- the last steps' sparse chains run after the last pixel kernel -- which a 20-step region pays in full and a
    # camera feed never does; the long region shows the steady state the schedule reaches.  Reported, never `value`.
    steady = None
    if not args.no_extras or os.environ.get("RMCV_BENCH_STEADY"):
        long_steps = 25 * args.steps
        barrier()
        t0 = time.perf_counter()
        for _ in range(long_steps):
            recs = step()
        barrier()
        dts = agree_max(time.perf_counter() - t0)
        steady = {"steps": long_steps, "ms_per_step": round(dts / long_steps * 1e3, 4), "frames_per_s": round(world * n * long_steps / dts, 1),
                  "note": "one region of 25 x --steps steps between barrier+synchronize pairs: the pipeline's fill and drain amortised; not the metric"}

    # ---- dev tool: RMCV_BENCH_AB="<option id>:<value A>:<value B>[:<pairs>]" -- the same loop, regions of 5 x --steps steps alternating between
    # two values of a context option (rmcv_ctx_set_option on every context), IN ONE PROCESS: the boxes drift by 2 % within a minute, which
    # A/B runs of separate processes cannot tell from an effect of 1 %.  Printed as `ab`; not the metric.
    ab = None
    if os.environ.get("RMCV_BENCH_AB"):
        f_ = os.environ["RMCV_BENCH_AB"].split(":")
        # "sched:<contexts>,<pixel streams>,<sparse streams>:<...>": the SHAPE of the schedule instead of a context option (start the
        # process with the larger of each: --streams / --pixel-streams / --sparse-streams, and GPU_MAX_HW_QUEUES to match)
        # "lib:<path of another build of librmcv_hip.so>": regions alternate between THIS build and that one (a second set of contexts on
        # the same frames; tools/build_variant*.sh make such builds) -- the only way to compare compile-time variants at better than +-3 %
        libab_ = f_[0] == "lib"
        if libab_:
            from rmcv_amd import abi as abi_
            lib_a, lib_b = abi_.lib(), abi_.load(os.path.abspath(f_[1]))
            abi_.use(lib_b)
            ctxs_b = [Context(device=local_rank, max_frames=n, max_width=W, max_height=H, max_contours=ctxs[0].limits.max_contours) for _ in range(n_sets)]
            for k, c in enumerate(ctxs_b):
                c.set_option(OPT_SPARSE_WAVES, int(os.environ.get("RMCV_SPARSE_WAVES", "4" if ns >= 3 else "8")))
                c.set_option(OPT_PIXEL_GROUPS, 2 if ns >= 2 else 3)
                c.bind_device_frames(frames_k[k].data_ptr(), n, H, W, keepalive=frames_k[k])
                if svm:
                    c.svm_load(*svm)
            abi_.use(lib_a)
            ctxs_a = list(ctxs)
            f_ = ["-2", "0", "1"] + f_[2:]
        # "stages:<mask>:<mask>": what the steps run (1 = pixel kernel only, 3 = + findContours, 7 = + fits, 15 = the whole path): what does each stage COST the step?
        stg_ = f_[0] == "stages"
        if stg_:
            masks_ = {0: int(f_[1]), 1: int(f_[2])}
            keep_stages_ = cur_stages[0]
            f_ = ["-4", "0", "1"] + f_[3:]
        sched_ = f_[0] == "sched"
        if sched_:
            shapes_ = {0: [int(x) for x in f_[1].split(",")], 1: [int(x) for x in f_[2].split(",")]}
            assert all(a <= b for sh in shapes_.values() for a, b in zip(sh, (ns, len(sAs), len(sBs))))
            f_ = ["-1", "0", "1"] + f_[3:]
        opt_, va_, vb_, pairs_ = int(f_[0]), int(f_[1]), int(f_[2]), int(f_[3]) if len(f_) > 3 else 12
        reg_ = 5 * args.steps
        res_ = {va_: [], vb_: []}
        for pr in range(pairs_):
            for v_ in ((va_, vb_) if pr % 2 == 0 else (vb_, va_)):
                if stg_:
                    barrier()
                    cur_stages[0] = masks_[v_] | (keep_stages_ & ~15)
                elif libab_:
                    barrier()
                    abi_.use(lib_b if v_ else lib_a)
                    ctxs[:] = ctxs_b if v_ else ctxs_a
                    for k_ in range(len(used)):
                        used[k_] = False                           # (the other set's events say nothing about this set's buffers)
                    step_no[0] = 0
                elif sched_:
                    barrier()
                    shape[:] = shapes_[v_]
                    step_no[0] = 0
                else:
                    for c in ctxs:
                        c.set_option(opt_, v_)
                if opt_ == OPT_HANDOVER:                           # the option AND the schedule that uses it (RMCV_STAGE_HANDOVER in step())
                    barrier()
                    ho[0] = bool(v_)
                for _ in range(2 * ns):
                    step()
                barrier()
                t0 = time.perf_counter()
                for _ in range(reg_):
                    step()
                barrier()
                res_[v_].append((time.perf_counter() - t0) / reg_ * 1e3)
        if stg_:
            barrier()
            cur_stages[0] = keep_stages_
        elif libab_:
            barrier()
            abi_.use(lib_a)
            ctxs[:] = ctxs_a
            for k_ in range(len(used)):
                used[k_] = False
            step_no[0] = 0
            for c in ctxs_b:
                c.close()
        elif sched_:
            barrier()
            shape[:] = [ns, len(sAs), len(sBs)]
            step_no[0] = 0
        else:
            for c in ctxs:
                c.set_option(opt_, va_)
        if opt_ == OPT_HANDOVER:
            barrier()
            ho[0] = bool(va_)
        ab = {"option": ("stage masks %d vs %d" % (masks_[0], masks_[1])) if stg_ else ("this build vs %s" % os.environ["RMCV_BENCH_AB"].split(":")[1]) if libab_ else ("sched %s vs %s" % (shapes_[0], shapes_[1])) if sched_ else opt_, "steps_per_region": reg_, "pairs": pairs_,
              "a": {"value": va_, "median_ms": round(float(np.median(res_[va_])), 4), "mean_ms": round(float(np.mean(res_[va_])), 4), "each": [round(x, 4) for x in res_[va_]]},
              "b": {"value": vb_, "median_ms": round(float(np.median(res_[vb_])), 4), "mean_ms": round(float(np.mean(res_[vb_])), 4), "each": [round(x, 4) for x in res_[vb_]]}}
        ab["b_over_a"] = round(ab["b"]["mean_ms"] / ab["a"]["mean_ms"], 4)

    # ---- what was computed (outside the timed region): status + gathered list sanity
    cnt = ctx.counts()
    bad = int(np.count_nonzero(cnt["status"] & 15))
    slow = int(np.count_nonzero(cnt["status"] & 16))              # frames findContours handed to the sequential scanner
    mid = int(np.count_nonzero(cnt["status"] & 64))               # frames beyond the LDS tables: mid tier (tables in global memory)
    n_arm_local = int(cnt["n_armours"].sum())
    gathered = None
    if rank == 0:
        arm, offs = rdist.unpack_records(recs, n, cap)
        gathered = int(arm.shape[0])
        assert offs[-1] == gathered and len(offs) == world * n + 1

    # ---- per-kernel durations with HIP events on the launch stream (same command, extra passes)
    stage = np.zeros(5)
    reps = max(5, min(args.steps, 20))
    for _ in range(reps):
        stage += np.asarray(nxt().run_timed(params, stages, sh))
    stage /= reps
    # one batch at a time, HIP events around the whole batch, median and min over >= 20 passes
    lone = []
    for c in ctxs:
        c.set_option(OPT_SPARSE_WAVES, 8)                          # the latency settings: a lone batch has the CUs to itself
        c.set_option(OPT_PIXEL_GROUPS, 3)
    for _ in range(max(20, reps)):
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cx = nxt()
        with torch.cuda.stream(stream):
            ea.record(stream)
            run_path(cx, stages, sh)
            cx.compact_armours_into(rec.data_ptr() + head, cap, rec.data_ptr(), sh)
            eb.record(stream)
        torch.cuda.synchronize()
        lone.append(ea.elapsed_time(eb))
    lone.sort()
    for c in ctxs:
        c.set_option(OPT_SPARSE_WAVES, int(os.environ.get("RMCV_SPARSE_WAVES", "4" if ns >= 3 else "8")))
        c.set_option(OPT_PIXEL_GROUPS, 2 if ns >= 2 else 3)
    fused_ms = None
    if legacy is None:          # what the steps actually launch: findContours + filter_lightblobs + filter_armours as one kernel
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            ctx.run(params, STAGE_ALL, sh)
            ea.record(stream)
            for _ in range(reps):
                ctx.run(params, STAGE_ALL & ~STAGE_BINARY, sh)
            eb.record(stream)
        torch.cuda.synchronize()
        fused_ms = ea.elapsed_time(eb) / reps
    if legacy is not None:      # run_timed drives the current API; time the legacy blob stage (k_match + k_pairs) on its own
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            ctx.run_legacy(legacy, params, STAGE_ALL, sh)
            ea.record(stream)
            for _ in range(reps):
                ctx.run_legacy(legacy, params, STAGE_BLOBS | STAGE_ARMOURS, sh)
            eb.record(stream)
        torch.cuda.synchronize()
        stage[4] += ea.elapsed_time(eb) / reps - stage[2] - stage[3]
        stage[2], stage[3] = ea.elapsed_time(eb) / reps, 0.0
    # the dominant kernel on its own: R back-to-back launches of k_binary between two HIP events recorded on the launch
    # stream, so the event/launch latency (~20 us, visible in stage_ms.binary) is amortised and the figure is the
    # kernel's duration, the same quantity rocprofv3 --kernel-trace reports
    R = 20
    def k_binary_alone(groups, rotate=True):
        for c in ctxs:
            c.set_option(OPT_PIXEL_GROUPS, groups)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            ctx.run(params, STAGE_BINARY, sh)
            e0.record(stream)
            for _ in range(R):
                # rotate: every launch reads frames, and writes buffers, last touched n_sets launches ago -- HBM, not the Infinity Cache
                (nxt() if rotate else ctx).run(params, STAGE_BINARY, sh)
            e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / R
    # alone the kernel runs at the library's default of 3 workgroups per CU (what `--streams 1` and a lone batch launch, and what the
    # serial rocprofv3 summary under profiles/ shows); the pipelined steps launch it with 2 per CU, two launches overlapping
    groups_in_steps = 2 if ns >= 2 else 3
    k1_ms = k_binary_alone(3)
    k1_steps_ms = k_binary_alone(groups_in_steps) if groups_in_steps != 3 else k1_ms
    k1_warm_ms = k_binary_alone(3, rotate=False)                    # rounds 1-2 and the first half of round 3 reported THIS as the roofline figure
    # the pixel kernels ALONE in the schedule the steps launch them in (two streams, the steps' workgroups per CU, every context's
    # own frames, no events): what the overlap of consecutive launches is worth (ramp and tail of one hidden behind the other)
    k1_pipe_ms = None
    if pipelined:
        torch.cuda.synchronize()
        for rep in range(2):
            t0p = time.perf_counter()
            for i in range(4 * R):
                ctxs[i % ns].run(params, STAGE_BINARY, sAs[i % len(sAs)].cuda_stream)
            torch.cuda.synchronize()
            k1_pipe_ms = (time.perf_counter() - t0p) / (4 * R) * 1e3
    for c in ctxs:
        c.set_option(OPT_PIXEL_GROUPS, groups_in_steps)
    achieved = n * BYTES_PER_FRAME / (k1_ms * 1e-3) / 1e9

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "k_binary_traffic_%s.json" % args.workload)   # per workload (frame size); c3's also under the old name
    if not os.path.exists(tpath):
        tpath = os.path.join(ROOT, "profiles", "k_binary_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("frames") == n and tj.get("width") == W and tj.get("height") == H:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "frames/sec (%dx%d BGR) armour detect" % (W, H), "value": round(value, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup + warm_steps, "ms_per_step": round(ms_per_step, 4),
        "timed_region": {"repeats": len(rep_dt), "ms_per_step_each": [round(d / args.steps * 1e3, 4) for d in rep_dt],
                         "ms_per_step_median": round(ms_per_step, 4), "ms_per_step_min": round(srt[0] / args.steps * 1e3, 4),
                         "value_at_min": round(world * n * args.steps / srt[0], 1), "warmup_steps_requested": args.warmup, "warmup_steps_by_time": warm_steps,
                         "host_enqueue_ms_per_step": round(sorted(enq_dt)[len(enq_dt) // 2] / args.steps * 1e3, 4),
                         "note": "each repeat = exactly `steps` steps between barrier+synchronize pairs; value/ms_per_step = the median repeat"},
        "steady_state": steady,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%s: batch=%d/GPU %dx%d BGR, blue lb=80, close3x3 + findContours + lightblob fit + armour "
                               "pairing%s%s" % (args.workload.upper(), n, W, H,
                                                (" + icon rectification + 7-class linear SVM (synthetic weights)" if svm else "") +
                                                (" [legacy blob stage: FindLightBlobs, minAreaRect boxes, camp vote]" if legacy else "") +
                                                (" + solve_PnP (IPPE square) and world position per armour" if args.pose else ""),
                                                " + RCCL gather of armour lists (C4)" if world > 1 else ""),
                   "frames_per_gpu": n, "stream_variant": args.variant, "parallelism": "frame-shard x%d" % world,
                   "double_buffered_steps": ns, "gpu_max_hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]), "pixel_groups_per_cu": 2 if ns >= 2 else 3, "sparse_waves_per_frame": int(os.environ.get("RMCV_SPARSE_WAVES", "4" if ns >= 3 else "8")), "schedule": ("software pipeline: pixel kernels alternate over %d streams, sparse stages on %d higher-priority streams, chained by events" % (len(sAs), len(sBs)) if pipelined else "alternating streams"),
                   "armours_rank0_shard": n_arm_local, "armours_gathered": gathered, "frames_over_capacity": bad,
                   "frames_slow_path": slow, "frames_mid_tier": mid,
                   "rccl_ranks": (dist.get_world_size() if use_dist else None), "gather": gather_note,
                   "frame_level_handover": handover},
        "lone_batch_ms": {"median": round(lone[len(lone) // 2], 4), "min": round(lone[0], 4), "passes": len(lone),
                          "note": "one batch at a time on one stream, events around detect + compaction (latency, not the metric)"},
        **({"ab": ab} if ab else {}),
        **({"ptrs": ["%x" % t.data_ptr() for t in frames_k]} if os.environ.get("RMCV_BENCH_PTRS") else {}),
        "path_hbm_frac": round(value / world * BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS, 4),
        "stage_ms": {"binary": round(float(stage[0]), 4), "contours": round(float(stage[1]), 4),
                     "blobs": round(float(stage[2]), 4), "armours": round(float(stage[3]), 4),
                     "sum": round(float(stage[4]), 4),
                     "note": "per-stage launches (rmcv_batch_run_timed); the steps run contours+blobs+armours as one fused "
                             "per-frame kernel: fused_sparse",
                     "fused_sparse": None if fused_ms is None else round(fused_ms, 4)},
        "roofline": {"kernel": "k_binary", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": n * BYTES_PER_FRAME, "avg_launch_ms": round(k1_ms, 4),
                     "launches_timed": R, "workgroups_per_cu": 3, "contexts_rotated": n_sets,
                     "same_frames_every_launch": {"avg_launch_ms": round(k1_warm_ms, 4),
                                                  "frac": round(n * BYTES_PER_FRAME / (k1_warm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                  "note": "NOT the roofline figure: 20 launches over the SAME gigabyte of frames, part of which the 256 MB Infinity "
                                                          "Cache still holds from the launch before (what this file reported as `roofline` until the second half "
                                                          "of round 3)"},
                     "as_launched_by_the_steps": {"workgroups_per_cu": groups_in_steps, "avg_launch_ms": round(k1_steps_ms, 4),
                                                  "frac": round(n * BYTES_PER_FRAME / (k1_steps_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                  "note": "alone, back to back; in the steps two such launches overlap (4 workgroups per CU resident)"},
                     "pixel_kernels_only_in_the_steps_schedule": None if k1_pipe_ms is None else {
                         "ms_per_launch": round(k1_pipe_ms, 4), "frac": round(n * BYTES_PER_FRAME / (k1_pipe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "%d launches alternating over the steps' %d pixel streams and %d contexts, nothing else on the machine, wall clock "
                                 "between two synchronisations: consecutive launches overlap, each hides the other's ramp and tail" % (4 * R, len(sAs), ns)}},
    }

    if not args.no_extras and rank == 0:
        # BASELINE config 2: red team, subtract + threshold + morphology only
        ex = {}
        for name, morph in (("dilate", MORPH_DILATE), ("close", MORPH_CLOSE)):
            p2 = default_params(camp=CAMP_RED, morph=morph)
            for _ in range(2):
                ctx.run(p2, STAGE_BINARY, sh)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                nxt().run(p2, STAGE_BINARY, sh)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t0
            ex[name + "_fps"] = round(n * args.steps / d2, 1)
        out["c2_binary_only"] = ex
        # detection only: the byte image `binary` is not written (RMCV_STAGE_NO_IMAGE; only the reference's debug view reads it,
        # executable/main.cpp:200-201).  NOT the metric: 3 B/px of algorithmic traffic instead of 4.
        if legacy is None and world == 1:       # (the loop below calls step(), which gathers: a collective only rank 0 entered would hang)
            cur_stages[0] = stages | STAGE_NO_IMAGE
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            d3 = (time.perf_counter() - t0) / args.steps
            cur_stages[0] = stages
            out["detect_only_no_image"] = {"fps": round(n / d3, 1), "ms_per_step": round(d3 * 1e3, 4), "bytes_per_frame": 3 * W * H,
                                           "hbm_frac": round(n * 3 * W * H / d3 / 1e9 / HBM_PEAK_GBS, 4),
                                           "note": "same armour lists; the 0/255 image is not materialised"}

    if args.density_sweep and world == 1:
        # ---- throughput against scene density (beside the metric): the same pipelined loop on the plain stream, the four dense
        # levels, and a plain batch with ONE dense4 frame in it (a camera frame with a lit window must not stall its launch)
        sweep = []
        region = max(100, 5 * args.steps)
        for label, var, one in [("plain", 0, False), ("dense1", 11, False), ("dense2", 12, False), ("dense3", 13, False),
                                ("dense4", 14, False), ("plain + one dense4 frame per batch", 0, True)]:
            if os.environ.get("RMCV_BENCH_SWEEP_LEVELS") and label.split()[0] not in os.environ["RMCV_BENCH_SWEEP_LEVELS"].split(",") \
                    and not (one and "one" in os.environ["RMCV_BENCH_SWEEP_LEVELS"].split(",")):
                continue                                           # dev knob (tools/ab_process_r3.sh one_dense): a subset of the levels
            for k in range(ns):
                hb = synth.batch(rank * n + k * 1000003, n, W, H, CAMP_BLUE, var, threads=nthreads)
                if one:
                    hb[n // 2] = synth.frame(rank * n + k * 1000003 + n // 2, W, H, CAMP_BLUE, 14)
                frames_k[k].copy_(torch.from_numpy(hb))
            torch.cuda.synchronize()
            for _ in range(20):
                step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(region):
                step()
            barrier()
            dsw = (time.perf_counter() - t0) / region
            st_ = ctx.counts()["status"]
            cnt_ = ctx.counts()
            sweep.append({"stream": label, "ms_per_step": round(dsw * 1e3, 4), "frames_per_s": round(n / dsw, 1),
                          "contours_per_frame": round(float(cnt_["n_contours"].mean()), 1),
                          "points_per_frame": round(float(cnt_["n_points"].mean()), 1),
                          "frames_mid_tier": int(np.count_nonzero(st_ & 64)), "frames_slow_path": int(np.count_nonzero(st_ & 16)),
                          "frames_over_capacity": int(np.count_nonzero(st_ & 15))})
        out["density_sweep"] = {"steps_per_region": region, "levels": sweep,
                                "note": "steady-state regions of the bench's own loop (%d batches in flight over %d sparse streams), rank-0 shard; not the metric" % (ns, len(sBs))}
        if not os.environ.get("RMCV_BENCH_SWEEP_LEVELS"):
            # The dense frame of a batch keeps ONE workgroup busy for 0.5-1 ms after the batch's other frames are through, and the
            # launches behind it on its sparse stream wait for it.  With a sparse stream PER batch in flight nothing is behind it:
            # the same two levels under that schedule, in a child process (the schedule is fixed when the streams are created).
            import subprocess
            env = dict(os.environ, RMCV_BENCH_SWEEP_LEVELS="plain,one", GPU_MAX_HW_QUEUES="12")
            cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--cpu-frames", "0",
                   "--no-extras", "--density-sweep", "--streams", "8", "--sparse-streams", "8", "--frames", str(n), "--workload", args.workload]
            try:
                cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                dj = json.loads(cp.stdout.strip().splitlines()[-1])
                out["density_sweep"]["deep_schedule"] = {
                    "batches_in_flight": 8, "sparse_streams": 8, "gpu_max_hw_queues": 12, "ms_per_step_20_step_regions": dj["ms_per_step"],
                    "levels": dj["density_sweep"]["levels"],
                    "note": "the same loop with 8 batches in flight and one sparse stream per batch (child process): one dense frame per batch no longer holds up the launches behind it"}
            except Exception as e:  # noqa: BLE001 -- a side measurement: report, never fail the bench line
                out["density_sweep"]["deep_schedule"] = {"error": repr(e)[:200]}
        for k in range(ns):                                        # back to the run's own stream for what follows
            frames_k[k].copy_(torch.from_numpy(host if k == 0 else synth.batch(rank * n + k * 1000003, n, W, H, CAMP_BLUE, args.variant, threads=nthreads)))
        torch.cuda.synchronize()

    if rank == 0 and world == 1 and not args.no_extras:
        # ---- the per-frame drop-in path (outside the timed region): the three C-ABI calls exactly as include/rmcv_shim.hpp issues
        # them for an unchanged executable/main.cpp:172-176, on ONE host frame at a time (pageable memory, as a cv::Mat is)
        import ctypes as C
        from rmcv_amd import OPT_FRAME_UPLOAD
        from rmcv_amd.abi import ARMOUR, LIGHTBLOB, POINT, lib, ptr
        L = lib()
        sf = {}
        c1 = Context(device=local_rank, max_frames=1, max_width=W, max_height=H)
        img = np.ascontiguousarray(host[0])
        binary = np.empty((H, W), np.uint8)
        pts, offs = np.empty(c1.limits.max_points, POINT), np.empty(c1.limits.max_contours + 1, np.int32)
        blobs, neg = np.empty(c1.limits.max_blobs, LIGHTBLOB), np.empty(c1.limits.max_contours, np.int32)
        arms = np.empty(c1.limits.max_armours, ARMOUR)
        nc, npt, nb, nn, na = (C.c_int32(0) for _ in range(5))

        def one_frame(want_binary=True):
            t = [time.perf_counter()]
            rc = L.rmcv_extract_color(c1._h, ptr(img), W, H, 3 * W, CAMP_BLUE, 80, MORPH_CLOSE, ptr(binary) if want_binary else None,
                                      ptr(pts), len(pts), ptr(offs), len(offs) - 1, C.byref(nc), C.byref(npt))
            t.append(time.perf_counter())
            rc |= L.rmcv_filter_lightblobs(c1._h, ptr(pts), ptr(offs), nc.value, C.c_float(70.0), C.c_float(1.5), C.c_float(80.0),
                                           C.c_double(10.0), C.c_double(99999.0), CAMP_BLUE, ptr(blobs), len(blobs), C.byref(nb), None,
                                           ptr(neg), C.byref(nn))
            t.append(time.perf_counter())
            rc |= L.rmcv_filter_armours(c1._h, ptr(blobs), nb.value, C.c_float(12.0), C.c_float(22.0), C.c_float(0.4), CAMP_BLUE,
                                        ptr(arms), len(arms), C.byref(na))
            t.append(time.perf_counter())
            assert rc == 0
            return [(t[i + 1] - t[i]) * 1e3 for i in range(3)]
        for mode, name in ((1, "pinned_staging"), (0, "runtime_pageable"), (2, "registered_in_place")):
            c1.set_option(OPT_FRAME_UPLOAD, mode)
            for _ in range(5):
                one_frame()
            m = np.array([one_frame() for _ in range(60)])
            tot = np.sort(m.sum(1))
            sf[name] = {"median_ms": round(float(tot[len(tot) // 2]), 4), "min_ms": round(float(tot[0]), 4),
                        "extract_color_ms": round(float(np.median(m[:, 0])), 4), "filter_lightblobs_ms": round(float(np.median(m[:, 1])), 4),
                        "filter_armours_ms": round(float(np.median(m[:, 2])), 4)}
        c1.set_option(OPT_FRAME_UPLOAD, 1)
        m = np.array([one_frame(False) for _ in range(60)])
        sf["pinned_staging_without_binary_image"] = {"median_ms": round(float(np.median(m.sum(1))), 4)}
        sf["armours"] = int(na.value)
        sf["note"] = ("one %dx%d host frame per call chain, results copied back to host after every call (what rm::extract_color / "
                      "filter_lightblobs / filter_armours return); 60 chains per mode; PCIe-inclusive, never `value`" % (W, H))
        c1.close()
        # the same chain from a C host (tools/frame_chain.c, compiled here if a C compiler is at hand): what an unchanged C++ caller
        # sees -- the figures above carry three ctypes calls per frame
        try:
            import shutil
            import subprocess
            import tempfile
            cc = shutil.which("gcc") or shutil.which("cc")
            if cc:
                exe = os.path.join(tempfile.mkdtemp(prefix="rmcv_fc_"), "frame_chain")
                libdir = os.path.join(ROOT, "rmcv_amd", "lib")
                subprocess.run([cc, "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "frame_chain.c"), "-o", exe,
                                "-L", libdir, "-lrmcv_hip", "-Wl,-rpath," + libdir], check=True, capture_output=True, timeout=120)
                cp = subprocess.run([exe, str(W), str(H)], capture_output=True, text=True, timeout=120)
                ch = {}
                for ln in cp.stdout.splitlines():
                    w_ = ln.split()
                    if len(w_) > 6 and w_[1] == "median":
                        ch[w_[0]] = {"median_ms": float(w_[2]), "min_ms": float(w_[4]), "p90_ms": float(w_[6])}
                if ch:
                    sf["c_host"] = dict(ch, note="tools/frame_chain.c: the three C-ABI calls from C, 300 chains per mode")
        except Exception as e:  # noqa: BLE001 -- a side measurement: report, never fail the bench line
            sf["c_host"] = {"error": repr(e)[:200]}
        out["single_frame_ms"] = sf

    if rank == 0 and world == 1 and args.cpu_frames > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O                                   # checker/baseline only, never the product path
        O.set_math_mode(0)
        m = min(args.cpu_frames, n)
        p = O.default_params()

        def cpu_frame(f):
            r = O.detect_frame(host[f], p)
            if legacy is not None:
                lb = O.find_lightblobs(host[f], r["pts"], r["offs"], 1.5, 80, 70, 10, 99999, False)[0]
                return O.filter_armours(lb, p)
            if svm:
                O.classify_armours(host[f], r["armours"], svm)
            if args.pose:
                O.locate_armours(r["armours"])
            return r["armours"]
        # bounded sample: whole passes over the first m frames until at least 5 s of CPU work (at most 16 passes)
        t0 = time.perf_counter()
        tot, passes = 0, 0
        while passes < 16 and (passes == 0 or time.perf_counter() - t0 < 5.0):
            tot = 0
            for f in range(m):
                tot += len(cpu_frame(f))
            passes += 1
        dc = time.perf_counter() - t0
        if "single_frame_ms" in out:
            ts = []
            for _ in range(30):
                t1 = time.perf_counter()
                cpu_frame(0)
                ts.append((time.perf_counter() - t1) * 1e3)
            out["single_frame_ms"]["cpu_port_median_ms"] = round(float(np.median(ts)), 4)
        try:
            cpu_model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
        except Exception:
            cpu_model = "unknown"
        out["cpu_baseline"] = {"value": round(passes * m / dc, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                               "cpu_model": cpu_model, "hardware_concurrency": os.cpu_count(),
                               "sample": "%d passes over the first %d frames of the same batch (%.1f s), oracle/ full path, 1 thread "
                                         "(the reference runs detection on one process_thread)" % (passes, m, dc),
                               "armours": tot}
        # the same port with every host core, frames in parallel (the C calls release the GIL)
        from concurrent.futures import ThreadPoolExecutor
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = min(cores, 16)          # the CPU share of a one-GPU box (the host shows all of its cores to every tenant)

        def one(f):
            return len(cpu_frame(f))
        t0 = time.perf_counter()
        passes_all = 0
        with ThreadPoolExecutor(cores) as ex:
            while passes_all < 64 and (passes_all == 0 or time.perf_counter() - t0 < 5.0):
                tot_all = sum(ex.map(one, range(n)))
                passes_all += 1
        dall = time.perf_counter() - t0
        out["cpu_baseline_all_cores"] = {"value": round(passes_all * n / dall, 2), "unit": "frames/s", "cores": cores, "kind": "port",
                                         "sample": "%d passes over all %d frames of the batch (%.1f s), one frame per task, %d threads (capped at "
                                                   "the 16-core share of a one-GPU box)" % (passes_all, n, dall, cores),
                                         "armours": tot_all}
        # ... and with EVERY hardware thread the host shows (SURVEY 8d asks for all cores; a one-GPU box's share is 16, so this
        # figure is what the whole host could do for one tenant, not what this tenant's quota sustains)
        every = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        if every > cores:
            t0 = time.perf_counter()
            passes_ev = 0
            with ThreadPoolExecutor(every) as ex:
                while passes_ev < 256 and (passes_ev == 0 or time.perf_counter() - t0 < 5.0):
                    tot_ev = sum(ex.map(one, range(n)))
                    passes_ev += 1
            dev_ = time.perf_counter() - t0
            out["cpu_baseline_every_hw_thread"] = {"value": round(passes_ev * n / dev_, 2), "unit": "frames/s", "cores": every, "kind": "port",
                                                   "sample": "%d passes over all %d frames of the batch (%.1f s), one frame per task, %d threads = "
                                                             "every hardware thread the host shows" % (passes_ev, n, dev_, every),
                                                   "armours": tot_ev}
    if rank == 0:
        print(json.dumps(out), flush=True)
    for g_ in (abi_gathers or []):
        g_.close()
    if use_dist:
        barrier()                                   # rank 0 has the extras and the CPU baseline to itself: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
