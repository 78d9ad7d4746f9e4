#!/usr/bin/env python3
"""bench.py -- frames/s of the armour-detection hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" = one pass of the hot path over one batch of 256 synthetic 1280x1024 BGR frames that are already resident in HBM:
rm::extract_color -> rm::filter_lightblobs -> rm::filter_armours (reference executable/main.cpp:172-176), the device-side
compaction of the armour lists, their copy to pinned host memory and -- for N > 1 -- the RCCL gather of the lists to rank 0
(BASELINE config 4).  Weak scaling: every rank owns its own 256 frames, no collective on the data path.

The schedule is the LIBRARY's: a step is ONE call, rmcv_pipeline_submit (include/rmcv_abi.h) -- eight batches in flight, pixel
kernels alternating over two HIP streams, the per-frame kernels on four higher-priority streams, chained by events inside
librmcv_hip.so.  tools/pipeline_bench.c drives the same calls from C; this file adds what the driver's contract asks for around it.

Prints ONE JSON line on rank 0: the contract fields plus
  roofline      k_binary (the kernel that moves the algorithmic 4 B/px), timed with HIP events on its own launch stream inside this
                process, COLD (every launch on another context's frames and buffers); `in_schedule_frac` = the same kernel in the
                steps' own schedule (two launches overlapping, nothing else on the machine)
  cpu_baseline  the CPU oracle (a port/restatement of the reference path, oracle/) timed on this box's host cores on a bounded
                sample of the same frames (rank 0, N=1 only)
  c5, density_sweep   short sub-records for BASELINE config 5 and for denser scenes (skipped by --no-extras)

Dev tool (--dev): RMCV_BENCH_AB="<option>:<a>:<b>" | "hot:<a>:<b>" | "sched:<depth,pix,sparse>:<...>" | "stages:<mask>:<mask>" | "lib:<another build>"
alternates two settings between regions of ONE process (two processes of one command differ by +-3 % on one box; regions inside a
process by 0.1 %).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read when the runtime starts.  The steps
# run 8 batches in flight over 2 pixel + 4 sparse streams; with the null stream and RCCL's that is more than 4 (a step's kernels
# that share a queue with another batch's cannot overlap: the gather of every step cost 15 % at 4).  12 (8 and 16 measure the same).
# librmcv_hip.so sets the same default when it is loaded; torch may start HIP before that, so it is set here as well.
# NOT more: with 16 or 24 queues and two pipelines' worth of streams in the process (the dev tool RMCV_BENCH_AB) every step took
# 0.63-0.67 ms instead of 0.257, and stayed there after the second pipeline was closed (tools/pipe_two.py): beyond ~12 active
# queues the hardware scheduler time-slices them.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

FRAMES = 256
WORKLOADS = {"c3": (1280, 1024), "c5": (1920, 1200),   # BASELINE.json configs[2] (the metric's config) and configs[4]
             "legacy": (1280, 1024)}                     # FindLightBlobs (minAreaRect boxes, camp vote) in place of filter_lightblobs
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
# environment knobs that change what the timed region runs: refused unless --dev, echoed in config.dev_knobs either way
DEV_KNOBS = ("RMCV_BENCH_STAGES", "RMCV_BENCH_AB", "RMCV_W4_ONE_PER_CU", "RMCV_EARLY_FREE", "RMCV_CHAIN_COLD", "RMCV_LAZY_BACK", "RMCV_SPARSE_WAVES", "RMCV_PIXEL_GROUPS", "RMCV_K1_BPC", "RMCV_FUSE_SPARSE", "RMCV_CONTOURS_LITERAL",
             "RMCV_K1_HALO_NT", "RMCV_K1_LINEAR", "RMCV_DENSE_DEFER", "RMCV_LIB_PATH", "RMCV_NO_MID", "RMCV_HOT_IDENTITY", "RMCV_WAIT_RUNTIME", "RMCV_HEAVY_PG", "RMCV_HEAVY_OFF", "RMCV_WS_ALWAYS", "RMCV_W4_ONE_PER_CU")
VARIANTS = {"plain": 0, "stress": 1, "dense1": 11, "dense2": 12, "dense3": 13, "dense4": 14, "dense": 14}


def variant_id(v):
    return VARIANTS[v] if v in VARIANTS else int(v)


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
    ap.add_argument("--one-dense", action="store_true", help="one dense4 frame in every batch of the stream (a camera frame with a lit window; a workload beside the metric)")
    ap.add_argument("--cpu-frames", type=int, default=256, help="frames per pass of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (C2, C5, density sweep, per-frame chain)")
    ap.add_argument("--density-sweep", action="store_true", help="the density sweep with all five levels instead of the default three")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3",
                    help="c3: 1280x1024 full path (the metric's config); c5: 1920x1200 full path + SVM digit classify on the icons; "
                         "legacy: c3 with rm::FindLightBlobs(fitEllipse=false) as the blob stage")
    ap.add_argument("--pose", action="store_true",
                    help="add the pose stage (rm::solve_PnP + world position per armour, SURVEY 8f-3) to every step")
    ap.add_argument("--streams", type=int, default=8,
                    help="batches in flight = depth of the pipeline (1 = strictly serial steps).  8 over 4 sparse streams since the end of "
                         "round 3: alternating regions of ONE process put it 4.3-4.5 %% ahead of round 2's 4 over 2 on C3 and 1 %% on C5")
    ap.add_argument("--pixel-streams", type=int, default=2)
    ap.add_argument("--sparse-streams", type=int, default=4)
    ap.add_argument("--dense-streams", type=int, default=0,
                    help="streams for the second, 8-wavefront launch that takes a batch's frames beyond findContours' LDS tables "
                         "(rmcv_pipeline_config::dense_streams; 0: the library's default, 2; -1: no such launch)")
    ap.add_argument("--hot-contexts", type=int, default=0,
                    help="rmcv_pipeline_config::hot_contexts: contexts the calm batches take turns at (0 = the library's default: derived from the bit planes' size, 4 for C3, 3 for C5; -1 = off)")
    ap.add_argument("--device-results", action="store_true",
                    help="leave the armour lists in HBM (rmcv_pipeline_config::host_results = 2) instead of copying every step's to pinned host memory")
    ap.add_argument("--gather", choices=("auto", "torch", "abi"), default="auto",
                    help="the armour-list gather of a launched run, through the pipeline's hook: torch.distributed.gather (asynchronous; the "
                         "default for more than one rank: rmcv_gather's multi-rank path has not run on hardware yet -- no multi-GPU box "
                         "was available to this build) or rmcv_gather, the C-ABI entry point that calls RCCL itself (what a C++ host "
                         "uses: rmcv_pipeline_set_gather; the default for a launched single rank, where it moves nothing)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the roofline kernel: warm-up, then 3 x 20 cold launches of k_binary (3 workgroups per CU, contexts and frame sets "
                         "in turn) between HIP events -- the command tools/profile_round4.sh runs under rocprofv3, so that the trace's average "
                         "duration of k_binary is ONE kind of launch (the full command mixes cold, warm, overlapping and 2-per-CU launches)")
    ap.add_argument("--dev", action="store_true", help="accept the environment knobs that change the timed region (%s)" % ", ".join(DEV_KNOBS))
    return ap.parse_args(argv)


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


def dev_knobs(environ, dev):
    """the environment knobs that are set; SystemExit when there are any and --dev was not given: a figure measured with one of
    them is not the metric, and the line must not look as if it were"""
    found = {k: environ[k] for k in DEV_KNOBS if environ.get(k)}
    if found and not dev:
        raise SystemExit("bench.py: %s set in the environment changes what the timed region runs; pass --dev to accept (the line then says so)"
                         % ", ".join(sorted(found)))
    return found


def c_host_chain(W, H, image_export=None):
    """tools/frame_chain.c compiled and run: the per-frame drop-in chain from a C host -> {mode: {median_ms, min_ms, p90_ms, extract_color_host_us}}"""
    try:
        import re
        import shutil
        import subprocess
        import tempfile
        cc = shutil.which("gcc") or shutil.which("cc")
        if not cc:
            return {"error": "no C compiler"}
        exe = os.path.join(tempfile.mkdtemp(prefix="rmcv_fc_"), "frame_chain")
        libdir = os.path.join(ROOT, "rmcv_amd", "lib")
        subprocess.run([cc, "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "frame_chain.c"), "-o", exe,
                        "-L", libdir, "-lrmcv_hip", "-Wl,-rpath," + libdir], check=True, capture_output=True, timeout=120)
        cp = subprocess.run([exe, str(W), str(H)], capture_output=True, text=True, timeout=120, env=dict(os.environ, **({} if image_export is None else {"RMCV_IMAGE_EXPORT": str(image_export), "RMCV_FRAME_UPLOAD": "0" if image_export == 0 else "1"})))
        ch, last_ = {}, None
        for ln in cp.stdout.splitlines():
            w_ = ln.split()
            if len(w_) > 6 and w_[1] == "median":
                last_ = w_[0]
                ch[last_] = {"median_ms": float(w_[2]), "min_ms": float(w_[4]), "p90_ms": float(w_[6])}
            elif last_ and ln.strip().startswith("extract_color on the host"):
                # where rmcv_extract_color's host time goes (rmcv_ctx_frame_timing): median (p90) microseconds per step
                ch[last_]["extract_color_host_us"] = {k.strip(): [float(a), float(b_)] for k, a, b_ in
                                                      re.findall(r"  ([a-zA-Z+ 2]+?) ([0-9.]+) \(([0-9.]+)\)", ln.split(":", 1)[1])}
        if not ch:
            return {"error": (cp.stderr or cp.stdout)[-200:]}
        return dict(ch, note="tools/frame_chain.c: the three C-ABI calls from C, 300 chains per mode")
    except Exception as e:  # noqa: BLE001 -- a side measurement: report, never fail the bench line
        return {"error": repr(e)[:200]}


def main():
    args = parse_args()
    world, rank, local_rank, launched = resolve_world(args, os.environ)
    knobs = dev_knobs(os.environ, args.dev)
    c_host_first = None
    if world == 1 and rank == 0 and not args.no_extras and not args.roofline_only and args.workload == "c3":
        c_host_first = c_host_chain(*WORKLOADS[args.workload])   # before this process touches the GPU (see single_frame_ms)
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

    from rmcv_amd import (CAMP_BLUE, CAMP_RED, MORPH_CLOSE, MORPH_DILATE, OPT_PIXEL_GROUPS, OPT_PIXEL_SHAPE, OPT_SPARSE_WAVES, STAGE_ALL, STAGE_BINARY,
                          STAGE_IDENTITY, STAGE_NO_IMAGE, STAGE_POSE, Context, LegacyParams, Pipeline, default_params, synth)
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

    def frame_sets(count, w, h, variant, one_dense=False):
        """`count` batches of n frames in HBM.  Set k of rank r starts at stream index r*n + k*1000003: steps that follow each other
        must not share input, or the later one would be served from the 256 MB Infinity Cache instead of HBM (k_binary alone, back
        to back: 0.215 ms on the same frames, 0.240 rotating over two sets, 0.2575 over four) -- which a camera feed never does."""
        out, first = [], None
        for k in range(count):
            hb = synth.batch(rank * n + k * 1000003, n, w, h, CAMP_BLUE, variant, threads=nthreads)
            if one_dense:
                hb[n // 2] = synth.frame(rank * n + k * 1000003 + n // 2, w, h, CAMP_BLUE, 14)
            if first is None:
                first = hb
            out.append(torch.from_numpy(hb).to(dev))              # resident in HBM before any timing
        return out, first

    # At least FOUR frame sets even when fewer batches are in flight; with 8 in flight, 8 (as round 3 measured it).
    n_sets = max(ns, 4)
    frames_k, host = frame_sets(n_sets, W, H, args.variant, args.one_dense)
    stages = STAGE_ALL | (STAGE_IDENTITY if args.workload == "c5" else 0) | (STAGE_POSE if args.pose else 0)
    if knobs.get("RMCV_BENCH_STAGES"):                            # dev knob: a partial path is NOT the metric
        stages = int(knobs["RMCV_BENCH_STAGES"])
    svm = synth.svm_weights() if args.workload == "c5" else None   # svm.xml is not in the reference: seeded stand-in weights
    params = default_params()                                     # main.cpp:172-176: BLUE, lb 80, close, ...
    legacy = LegacyParams(1.5, 80, 70, 10, 99999, int(os.environ.get("RMCV_LEGACY_FIT", "0"))) if args.workload == "legacy" else None
    max_contours = 4096 if args.variant >= 10 or args.one_dense else 2048   # (the dense streams have up to ~2100 contours per frame)

    def make_pipeline(depth, pix, sp, dense=0, w=W, h=H, mc=max_contours, with_svm=svm, host_results=None):
        # a lone batch has the CUs to itself: 8 wavefronts per frame and 3 pixel workgroups per CU; batches in flight share every CU:
        # 4 and 2 (the library's own rule, rmcv_pipeline_create; the dev knobs override it)
        pl = Pipeline(device=local_rank, depth=depth, pixel_streams=pix, sparse_streams=sp, armour_cap=n * 8,
                      sparse_waves=int(knobs.get("RMCV_SPARSE_WAVES", 0)), pixel_groups=int(knobs.get("RMCV_PIXEL_GROUPS", 0)),
                      host_results=host_results or (2 if args.device_results else 1),
                      dense_streams=dense or args.dense_streams, hot_contexts=args.hot_contexts,
                      max_frames=n, max_width=w, max_height=h, max_contours=mc)
        for c in pl.contexts:
            if with_svm:
                c.svm_load(*with_svm)
            if args.pose:
                c.pnp_load()                                      # camera constants of executable/main.cpp:7-19
        return pl

    pl = make_pipeline(ns, args.pixel_streams, args.sparse_streams)
    info = pl.info
    cap = info.armour_cap
    head = info.armours_offset

    # ---- the gather of a launched run rides on the pipeline's hook
    if args.gather == "auto":
        args.gather = "abi" if world == 1 else "torch"
    gather_note, hook, abi_gather = (args.gather if use_dist else None), None, None
    if use_dist and args.gather == "abi":
        abi_gather = rdist.AbiGather(local_rank)
        # self-check before anything is timed: the communicator moves a stamped record from every rank to its slot on the root
        probe = torch.full((4096,), rank + 1, dtype=torch.uint8, device=dev)
        rb = abi_gather.new_recv(probe)
        parts = abi_gather.gather(probe, rb, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ok = 1
        if rank == 0 and any(int(pt.min()) != r_ + 1 or int(pt.max()) != r_ + 1 for r_, pt in enumerate(parts)):
            ok = 0
        t_ok = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.broadcast(t_ok, src=0)
        if int(t_ok.item()) != 1:          # never time a path that moved wrong bytes: fall back to torch.distributed.gather, and say so
            abi_gather.close()
            abi_gather, gather_note = None, "torch (rmcv_gather failed its self-check)"
        else:
            pl.set_gather(abi_gather._h, 0)
    if use_dist and abi_gather is None:
        hook = rdist.TorchGatherHook(info.record_bytes, ns, dev)
        pl.set_hook(hook)

    # the kernel that moves the algorithmic 4 B/px in the steps: k_binary_ws in the contexts that take turns while the batches are calm
    # (rmcv_pipeline_config::hot_contexts), k_binary where that is off or does not apply (a pose stage, the legacy blob stage)
    hot_mode = info.hot_contexts > 0 and legacy is None and not (stages & (STAGE_IDENTITY | STAGE_POSE))
    roof_kernel = "k_binary_ws" if hot_mode else "k_binary"
    roof_ctxs = pl.contexts[:info.hot_contexts] if hot_mode else pl.contexts

    if args.roofline_only:
        ctxs = roof_ctxs
        for c in ctxs:
            c.set_option(OPT_PIXEL_GROUPS, 3)
            c.set_option(OPT_PIXEL_SHAPE, 1 if hot_mode else 0)
        stream = torch.cuda.Stream(device=dev)
        sh, R, each = stream.cuda_stream, 20, []
        for rep in range(4):                                         # the first repeat is the warm-up (clocks, code objects)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            with torch.cuda.stream(stream):
                e0.record(stream)
                for i in range(R):
                    c = ctxs[(rep * R + i) % len(ctxs)]
                    c.bind_device_frames(frames_k[(rep * R + i) % n_sets].data_ptr(), n, H, W)
                    c.run(params, STAGE_BINARY, sh)
                e1.record(stream)
            torch.cuda.synchronize()
            each.append(e0.elapsed_time(e1) / R)
        ms = sorted(each[1:])[1]
        print(json.dumps({"roofline_only": True, "kernel": roof_kernel, "workgroups_per_cu": 1 if hot_mode else 3, "contexts_rotated": len(ctxs), "frame_sets_rotated": n_sets,
                          "launches": 4 * R, "avg_launch_ms_each_repeat": [round(x, 4) for x in each],
                          "avg_launch_ms": round(ms, 4), "achieved_GBps": round(n * BYTES_PER_FRAME / (ms * 1e-3) / 1e9, 1),
                          "frac": round(n * BYTES_PER_FRAME / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": n * BYTES_PER_FRAME,
                          "note": "cold: every launch reads another frame set; its context's buffers were last written %d launches ago (the contexts the "
                                  "steps' batches take turns at); one stream, HIP events around each 20 launches" % len(ctxs)}), flush=True)
        pl.close()
        return

    cur = {"pl": pl, "stages": stages, "sets": frames_k, "w": W, "h": H, "max_submit": 0.0}
    step_no = [0]

    def step():
        """ONE call into the library: the next batch, on the next frame set"""
        fr = cur["sets"][step_no[0] % len(cur["sets"])]
        step_no[0] += 1
        t_s = time.perf_counter()
        t_ = cur["pl"].submit(fr.data_ptr(), n, cur["h"], cur["w"], params, cur["stages"], legacy=legacy)
        t_s = time.perf_counter() - t_s
        if t_s > cur["max_submit"]:
            cur["max_submit"] = t_s                            # the longest single submit call (host time): a submit never blocks
        return t_

    def barrier():
        # the pipeline and every gather are drained BEFORE the process group's own collectives (barrier, all_reduce) are enqueued:
        # rmcv_gather's communicator and torch's never have kernels resident together
        cur["pl"].drain()
        if hook is not None:
            hook.wait_all()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def agree_max(x):
        """the same number on every rank (MAX): ranks must take the same decisions, a step contains a collective"""
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    hot_each = []                                              # batches of each region that ran in a hot context
    submit_max_each = []                                       # the longest single submit call of each region (host seconds), as Python's clock sees it
    submit_max_lib_each = []                                   # ... as the library's own clock sees it (rmcv_pipeline_info::max_submit_us: no ctypes, no GC pause)

    def regions(steps, repeats):
        """`repeats` regions of exactly `steps` steps between barrier + synchronize pairs -> (wall seconds each, host enqueue seconds each)"""
        rep, enq = [], []
        for _ in range(repeats):
            barrier()
            h0 = cur["pl"].get_info().hot_batches
            cur["max_submit"] = 0.0
            cur["pl"]._lib.rmcv_pipeline_reset_stats(cur["pl"]._h)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            enq.append(time.perf_counter() - t0)       # host time to enqueue the region's steps (the GPU may still be running them)
            barrier()
            rep.append(agree_max(time.perf_counter() - t0))
            hot_each.append(int(cur["pl"].get_info().hot_batches - h0))
            submit_max_each.append(cur["max_submit"])
            submit_max_lib_each.append(cur["pl"].get_info().max_submit_us * 1e-6)
        return rep, enq

    def median(x):
        s = sorted(x)
        return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])

    if knobs.get("RMCV_BENCH_STAGES"):                           # dev knob: later stages need the planes of a full pass
        cur["stages"] = STAGE_ALL
        for _ in range(ns):
            step()
        barrier()
        cur["stages"] = stages
    for _ in range(args.warmup):
        step()
    barrier()
    # warm-up by time as well: a GPU that idled while the frames were generated has to ramp its clocks; 5 steps are 1.5 ms
    tw, warm_steps = time.perf_counter(), 0
    while agree_max(time.perf_counter() - tw) < args.warmup_seconds:
        for _ in range(max(1, args.steps)):
            step()
        warm_steps += max(1, args.steps)
        barrier()
    # the timed region: EXACTLY --steps steps between two (barrier + synchronize), MAX over ranks; repeated --repeats times,
    # value = the median repeat (median and min over the passes)
    del hot_each[:]
    del submit_max_each[:]
    del submit_max_lib_each[:]
    rep_dt, enq_dt = regions(args.steps, max(1, args.repeats))
    hot_timed = list(hot_each)
    submit_max_timed = list(submit_max_lib_each)
    if hot_mode and 2 * sum(hot_timed) < len(hot_timed) * args.steps:   # (a stream with dense frames is never calm: its steps run k_binary)
        hot_mode, roof_kernel, roof_ctxs = False, "k_binary", pl.contexts
    dt = median(rep_dt)
    ms_per_step = dt / args.steps * 1e3
    value = world * n * args.steps / dt
    # beside the metric: ONE long region (25 x --steps steps): a timed region starts with an empty pipeline and ends by draining it,
    # which a 20-step region pays in full and a camera feed never does.  Reported, never `value`.
    steady = None
    collect_variant = None
    if not args.no_extras:
        # beside the metric: the same regions with every list READ by the host inside the loop -- rmcv_pipeline_collect of ticket
        # t - depth + 1 right behind submit t, as the header's example loop does (the metric's region only submits; its lists go to
        # pinned host memory every step but nobody reads them until the region is over)
        def step_collect():
            t_ = step()
            if t_ + 1 >= base_t + ns:
                cur["pl"].collect(t_ - ns + 1)
        reps_c = []
        for _ in range(3):
            barrier()
            base_t = cur["pl"].get_info().submitted
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_collect()
            barrier()
            reps_c.append(agree_max(time.perf_counter() - t0))
        dc = median(reps_c)
        collect_variant = {"ms_per_step": round(dc / args.steps * 1e3, 4), "frames_per_s": round(world * n * args.steps / dc, 1),
                           "ms_per_step_each": [round(x / args.steps * 1e3, 4) for x in reps_c],
                           "note": "the timed region's loop with rmcv_pipeline_collect(t - depth + 1) behind every submit: the host reads every list as it goes; not the metric"}
        long_steps = 25 * args.steps
        (dts,), _ = regions(long_steps, 1)
        steady = {"steps": long_steps, "ms_per_step": round(dts / long_steps * 1e3, 4), "frames_per_s": round(world * n * long_steps / dts, 1),
                  "note": "one region of 25 x --steps steps between barrier+synchronize pairs: the pipeline's fill and drain amortised; not the metric"}

    # ---- what was computed (outside the timed region): the last `depth` batches through rmcv_pipeline_collect + the gathered list
    barrier()
    arm_by_set = {}
    submitted = pl.get_info().submitted                          # tickets count the submits on this pipeline = the steps so far
    for t_ in range(max(0, submitted - ns), submitted):
        a_, o_ = pl.collect(t_)
        arm_by_set[t_ % n_sets] = int(len(a_))                    # (step t ran on frame set t % n_sets)
        assert o_[-1] == len(a_)
    last_ticket = submitted - 1
    ctx_last = pl.context_of(last_ticket)
    cnt = ctx_last.counts()
    bad = int(np.count_nonzero(cnt["status"] & 15))
    slow = int(np.count_nonzero(cnt["status"] & 16))              # frames findContours handed to the sequential scanner
    mid = int(np.count_nonzero(cnt["status"] & 64))               # frames beyond the LDS tables: mid tier (tables in global memory)
    n_arm_local = arm_by_set.get(0, int(cnt["n_armours"].sum()))   # the armours of frame set 0 (= the frames the CPU baseline runs on)
    gathered = None
    if rank == 0:
        if use_dist and hook is not None:
            recs = hook.records(last_ticket)
        elif use_dist:
            d_, b_ = pl.gathered(last_ticket)
            whole = rdist.tensor_at(d_, b_, dev)
            recs = [whole[r_ * info.record_bytes:(r_ + 1) * info.record_bytes] for r_ in range(world)]
        else:
            d_, _s = pl.record(last_ticket)
            recs = [rdist.tensor_at(d_, info.record_bytes, dev)]
        arm, offs = rdist.unpack_records(recs, n, cap)
        gathered = int(arm.shape[0])
        assert offs[-1] == gathered and len(offs) == world * n + 1

    # ---- dev tool: RMCV_BENCH_AB -- regions of 5 x --steps steps alternating between two settings IN ONE PROCESS
    ab = None
    if knobs.get("RMCV_BENCH_AB"):
        f_ = knobs["RMCV_BENCH_AB"].split(":")
        kind = f_[0]
        pairs_ = 12
        pls, stg, opt_, hot_ = {0: pl, 1: pl}, {0: stages, 1: stages}, None, None
        # A is always the run's own pipeline, B a second one (at most two rings, i.e. 12 streams, exist at a time)
        if kind == "lib":        # "lib:<path>": THIS build against another build of librmcv_hip.so (tools/build_variant*.sh)
            from rmcv_amd import abi as abi_
            lib_a, lib_b = abi_.lib(), abi_.load(os.path.abspath(f_[1]))
            abi_.use(lib_b)
            pls[1] = make_pipeline(ns, args.pixel_streams, args.sparse_streams)
            abi_.use(lib_a)
            label, pairs_ = "this build vs %s" % f_[1], int(f_[2]) if len(f_) > 2 else 12
        elif kind == "sched":    # "sched:<depth,pix,sparse[,dense]>": the run's schedule against another shape
            sh = [int(x) for x in f_[1].split(",")]
            pls[1] = make_pipeline(*sh)
            label, pairs_ = "sched %s vs %s" % ([ns, args.pixel_streams, args.sparse_streams], sh), int(f_[2]) if len(f_) > 2 else 12
        elif kind == "env":      # "env:<NAME>:<b>": against a pipeline created under another value of an environment knob of the library
            before = os.environ.get(f_[1])
            os.environ[f_[1]] = f_[2]
            pls[1] = make_pipeline(ns, args.pixel_streams, args.sparse_streams)
            if before is None:
                os.environ.pop(f_[1])
            else:
                os.environ[f_[1]] = before
            label, pairs_ = "%s=%s vs %s" % (f_[1], before, f_[2]), int(f_[3]) if len(f_) > 3 else 12
        elif kind == "hostres":  # "hostres:<b>": rmcv_pipeline_config::host_results (1: lists copied to pinned memory every step, 2: left in HBM)
            pls[1] = make_pipeline(ns, args.pixel_streams, args.sparse_streams, host_results=int(f_[1]))
            label, pairs_ = "host_results %d vs %s" % (info.host_results, f_[1]), int(f_[2]) if len(f_) > 2 else 12
        elif kind == "hot":      # "hot:<a>:<b>": rmcv_pipeline_config::hot_contexts, switched on the run's own pipeline (0 = off)
            hot_ = {0: int(f_[1]), 1: int(f_[2])}
            label, pairs_ = "hot_contexts %s vs %s" % (f_[1], f_[2]), int(f_[3]) if len(f_) > 3 else 12
        elif kind == "stages":   # "stages:<mask>:<mask>": what each stage COSTS the step (1 = pixel kernel only, 3 = + findContours, ...)
            stg = {0: int(f_[1]) | (stages & ~15), 1: int(f_[2]) | (stages & ~15)}
            label, pairs_ = "stage masks %s vs %s" % (f_[1], f_[2]), int(f_[3]) if len(f_) > 3 else 12
        else:                    # "<option id>:<a>:<b>": rmcv_ctx_set_option on every context of the ring
            opt_, vals = int(f_[0]), {0: int(f_[1]), 1: int(f_[2])}
            label, pairs_ = "option %d: %d vs %d" % (opt_, vals[0], vals[1]), int(f_[3]) if len(f_) > 3 else 12
        reg_, res_ = 5 * args.steps, {0: [], 1: []}
        for pr in range(pairs_):
            for v_ in ((0, 1) if pr % 2 == 0 else (1, 0)):
                barrier()
                cur["pl"], cur["stages"] = pls[v_], stg[v_]
                if opt_ is not None:
                    for c in pl.contexts:
                        c.set_option(opt_, vals[v_])
                if hot_ is not None:
                    pl.set_hot_contexts(hot_[v_])
                for _ in range(2 * ns):
                    step()
                (d_,), _ = regions(reg_, 1)
                res_[v_].append(d_ / reg_ * 1e3)
        barrier()
        cur["pl"], cur["stages"] = pl, stages
        if hot_ is not None:
            pl.set_hot_contexts(info.hot_contexts)
        if opt_ is not None:
            for c in pl.contexts:
                c.set_option(opt_, vals[0])
        for v_ in (0, 1):
            if pls[v_] is not pl:
                pls[v_].close()
        ab = {"what": label, "steps_per_region": reg_, "pairs": pairs_,
              "a": {"median_ms": round(float(np.median(res_[0])), 4), "mean_ms": round(float(np.mean(res_[0])), 4), "each": [round(x, 4) for x in res_[0]]},
              "b": {"median_ms": round(float(np.median(res_[1])), 4), "mean_ms": round(float(np.mean(res_[1])), 4), "each": [round(x, 4) for x in res_[1]]}}
        ab["b_over_a"] = round(ab["b"]["mean_ms"] / ab["a"]["mean_ms"], 4)

    # ---- "alone" measurements on the ring's own contexts (the pipeline is drained), each on its own frames: HBM, not the Infinity Cache
    barrier()
    ctxs = pl.contexts
    for k, c in enumerate(ctxs):
        c.bind_device_frames(frames_k[k % n_sets].data_ptr(), n, H, W)
    rot = [0]

    def nxt():
        """the next context in turn: its frames were last read, and its buffers last written, a ring ago"""
        rot[0] = (rot[0] + 1) % len(ctxs)
        return ctxs[rot[0]]

    def run_path(c, st, hs):
        if legacy is not None:
            c.run_legacy(legacy, params, st, hs)
        else:
            c.run(params, st, hs)
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream
    rec = rdist.new_record(n, cap, dev)
    groups_in_steps, waves_in_steps = info.pixel_groups, info.sparse_waves

    def settings(waves, groups):
        for c in ctxs:
            c.set_option(OPT_SPARSE_WAVES, waves)
            c.set_option(OPT_PIXEL_GROUPS, groups)
    # per-kernel durations with HIP events on the launch stream
    stage = np.zeros(5)
    reps = max(5, min(args.steps, 20))
    for _ in range(reps):
        stage += np.asarray(nxt().run_timed(params, stages, sh))
    stage /= reps
    # one batch at a time, HIP events around the whole batch, median and min over >= 20 passes -- at the latency settings
    settings(8, 3)
    lone = []
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
    fused_ms = None
    if legacy is None:          # what the steps actually launch: findContours + filter_lightblobs + filter_armours as one kernel
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            ctxs[0].run(params, STAGE_ALL, sh)
            ea.record(stream)
            for _ in range(reps):
                ctxs[0].run(params, STAGE_ALL & ~STAGE_BINARY, sh)
            eb.record(stream)
        torch.cuda.synchronize()
        fused_ms = ea.elapsed_time(eb) / reps
    # the dominant kernel on its own: R back-to-back launches of k_binary between two HIP events recorded on the launch stream (the
    # event/launch latency amortised: the figure is the kernel's duration, the quantity rocprofv3 --kernel-trace reports)
    R = 20

    rot_r = [0]

    def k_binary_alone(groups, rotate=True, shape=0):
        """R launches back to back; cold: each reads another frame set, in the contexts the steps' batches take turns at"""
        for c in ctxs:
            c.set_option(OPT_PIXEL_GROUPS, groups)
            c.set_option(OPT_PIXEL_SHAPE, shape)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            ctxs[0].run(params, STAGE_BINARY, sh)
            e0.record(stream)
            for _ in range(R):
                if rotate:
                    rot_r[0] += 1
                    c = roof_ctxs[rot_r[0] % len(roof_ctxs)]
                    c.bind_device_frames(frames_k[rot_r[0] % n_sets].data_ptr(), n, H, W)
                else:
                    c = ctxs[0]
                c.run(params, STAGE_BINARY, sh)
            e1.record(stream)
        torch.cuda.synchronize()
        for c in ctxs:
            c.set_option(OPT_PIXEL_SHAPE, 1)                      # (the contexts' default)
        return e0.elapsed_time(e1) / R
    k1_ms = k_binary_alone(3, shape=1 if hot_mode else 0)          # the steps' kernel (k_binary: the library's default of 3 workgroups per CU)
    k1_steps_ms = k_binary_alone(groups_in_steps) if (groups_in_steps != 3 or hot_mode) else k1_ms   # k_binary as the steps' other batches launch it
    k1_warm_ms = k_binary_alone(3, rotate=False, shape=1 if hot_mode else 0)
    # the pixel kernels ALONE in the schedule the steps launch them in (the steps' streams, workgroups per CU and frame sets, no sparse
    # stage): what the overlap of consecutive launches is worth -- each hides the other's ramp and tail
    settings(waves_in_steps, groups_in_steps)
    k1_pipe_ms = None
    if ns > 1:
        cur["stages"] = stages & (STAGE_BINARY | STAGE_NO_IMAGE)
        if not hot_mode:
            pl.set_hot_contexts(0)                                # (the pixel kernel the steps run, not the one a pixel-only mask would get)
        for _ in range(ns):
            step()
        (d_,), _ = regions(4 * R, 1)
        k1_pipe_ms = d_ / (4 * R) * 1e3
        pl.set_hot_contexts(info.hot_contexts)
        cur["stages"] = STAGE_ALL | (stages & ~15)              # (leave full lists behind in every slot)
        for _ in range(ns):
            step()
        barrier()
        cur["stages"] = stages
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

    def frac(ms):
        return round(n * BYTES_PER_FRAME / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    srt = sorted(rep_dt)
    out = {
        "metric": "frames/sec (%dx%d BGR) armour detect" % (W, H), "value": round(value, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "timed_region": {"repeats": len(rep_dt), "ms_per_step_each": [round(d / args.steps * 1e3, 4) for d in rep_dt],
                         "batches_in_hot_contexts_each": hot_timed,
                         "ms_per_step_median": round(ms_per_step, 4), "ms_per_step_min": round(srt[0] / args.steps * 1e3, 4),
                         "value_at_min": round(world * n * args.steps / srt[0], 1), "warmup_steps_requested": args.warmup, "warmup_steps_by_time": warm_steps,
                         "host_enqueue_ms_per_step": round(median(enq_dt) / args.steps * 1e3, 4),
                         "max_submit_host_ms_each": [round(x * 1e3, 4) for x in submit_max_timed],
                         "host_blocking_calls": int(pl.get_info().host_blocking_calls),
                         "note": "each repeat = exactly `steps` calls of rmcv_pipeline_submit between barrier+synchronize pairs; value/ms_per_step = the median repeat. "
                                 "A region starts on an empty machine and ends with a drain: its second pixel launch is held back (k_delay, <= 60 us) so that the first takes "
                                 "every CU, and its last batch's sparse stage runs with the latency kernel (the drain knows nothing follows) -- both are the library's "
                                 "behaviour for any burst, not the bench's; steady_state and steps_with_collect show the same loop without / beside them"},
        "steady_state": steady, "steps_with_collect": collect_variant,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%s: batch=%d/GPU %dx%d BGR, blue lb=80, close3x3 + findContours + lightblob fit + armour "
                               "pairing%s%s" % (args.workload.upper(), n, W, H,
                                                (" + icon rectification + 7-class linear SVM (synthetic weights)" if svm else "") +
                                                (" [legacy blob stage: FindLightBlobs, minAreaRect boxes, camp vote]" if legacy else "") +
                                                (" + solve_PnP (IPPE square) and world position per armour" if args.pose else ""),
                                                " + RCCL gather of armour lists (C4)" if world > 1 else ""),
                   "frames_per_gpu": n, "stream_variant": args.variant, "one_dense_frame_per_batch": args.one_dense, "parallelism": "frame-shard x%d" % world,
                   "host_api": "rmcv_pipeline_submit (librmcv_hip.so): one call per step",
                   "batches_in_flight": info.depth, "pixel_streams": info.pixel_streams, "sparse_streams": info.sparse_streams, "dense_streams": info.dense_streams,
                   "hot_contexts": info.hot_contexts, "batches_in_hot_contexts": int(pl.get_info().hot_batches), "batches_finished_with_the_latency_kernel": int(pl.get_info().latency_batches), "pixel_kernel_of_calm_batches": roof_kernel,
                   "frame_sets": n_sets, "gpu_max_hw_queues": info.hw_queues_env, "pixel_groups_per_cu": groups_in_steps,
                   "sparse_waves_per_frame": waves_in_steps, "results_to_host_every_step": info.host_results == 1,
                   "stages": stages, "dev_knobs": knobs or None,
                   "armours_rank0_shard": n_arm_local, "armours_by_frame_set": [arm_by_set.get(k) for k in range(n_sets)], "armours_gathered": gathered,
                   "frames_over_capacity": bad, "frames_slow_path": slow, "frames_mid_tier": mid, "batches_with_dense_frames_split_off": int(pl.get_info().dense_split),
                   "rccl_ranks": (dist.get_world_size() if use_dist else None), "gather": gather_note},
        "lone_batch_ms": {"median": round(lone[len(lone) // 2], 4), "min": round(lone[0], 4), "passes": len(lone),
                          "note": "one batch at a time on one stream, events around detect + compaction (latency, not the metric)"},
        **({"ab": ab} if ab else {}),
        "path_hbm_frac": round(value / world * BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS, 4),
        "stage_ms": {"binary": round(float(stage[0]), 4), "contours": round(float(stage[1]), 4),
                     "blobs": round(float(stage[2]), 4), "armours": round(float(stage[3]), 4),
                     "sum": round(float(stage[4]), 4),
                     "note": "per-stage launches (rmcv_batch_run_timed); the steps run contours+blobs+armours as one fused "
                             "per-frame kernel: fused_sparse",
                     "fused_sparse": None if fused_ms is None else round(fused_ms, 4)},
        "roofline": {"kernel": roof_kernel, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "in_schedule_frac": None if k1_pipe_ms is None else frac(k1_pipe_ms),
                     "algorithmic_bytes_per_launch": n * BYTES_PER_FRAME, "avg_launch_ms": round(k1_ms, 4),
                     "launches_timed": R, "workgroups_per_cu": 1 if hot_mode else 3, "contexts_rotated": len(roof_ctxs), "frame_sets_rotated": n_sets,
                     "note": "frac: one launch at a time, cold (every launch reads another frame set, in the contexts the steps' batches take turns at); in_schedule_frac: the same "
                             "kernel as the steps launch it -- two launches overlapping on two streams, each hiding the other's ramp and tail",
                     "same_frames_every_launch": {"avg_launch_ms": round(k1_warm_ms, 4), "frac": frac(k1_warm_ms),
                                                  "note": "NOT the roofline figure: 20 launches over the SAME gigabyte of frames, part of which the 256 MB "
                                                          "Infinity Cache still holds from the launch before"},
                     "k_binary_as_the_other_batches_launch_it": {"workgroups_per_cu": groups_in_steps, "avg_launch_ms": round(k1_steps_ms, 4), "frac": frac(k1_steps_ms),
                                                  "note": "k_binary alone, back to back (batches with dense frames, a classifier or pose stage: two such launches overlap, 4 workgroups per CU resident)"},
                     "pixel_kernels_only_in_the_steps_schedule": None if k1_pipe_ms is None else {
                         "ms_per_launch": round(k1_pipe_ms, 4), "frac": frac(k1_pipe_ms),
                         "note": "%d steps of the pipeline with the stage mask cut down to RMCV_STAGE_BINARY, wall clock between two drains" % (4 * R)}},
    }

    extras = not args.no_extras and rank == 0 and world == 1
    if not args.no_extras and rank == 0:
        # BASELINE config 2: red team, subtract + threshold + morphology only
        ex = {}
        for name, morph in (("dilate", MORPH_DILATE), ("close", MORPH_CLOSE)):
            p2 = default_params(camp=CAMP_RED, morph=morph)
            for _ in range(2):
                ctxs[0].run(p2, STAGE_BINARY, sh)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                nxt().run(p2, STAGE_BINARY, sh)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t0
            ex[name + "_fps"] = round(n * args.steps / d2, 1)
        out["c2_binary_only"] = ex
    if extras and legacy is None:
        # detection only: the byte image `binary` is not written (RMCV_STAGE_NO_IMAGE; only the reference's debug view reads it,
        # executable/main.cpp:200-201).  NOT the metric: 3 B/px of algorithmic traffic instead of 4.
        cur["stages"] = stages | STAGE_NO_IMAGE
        for _ in range(ns):
            step()
        (d3,), _ = regions(args.steps, 1)
        d3 /= args.steps
        cur["stages"] = stages
        out["detect_only_no_image"] = {"fps": round(n / d3, 1), "ms_per_step": round(d3 * 1e3, 4), "bytes_per_frame": 3 * W * H,
                                       "hbm_frac": round(n * 3 * W * H / d3 / 1e9 / HBM_PEAK_GBS, 4),
                                       "note": "same armour lists; the 0/255 image is not materialised"}

    if extras and args.workload == "c3" and args.variant == 0 and not args.one_dense:
        # ---- throughput against scene density (beside the metric), the steps' own loop and schedule: the plain stream, dense levels
        # (up to +2000 specks and 13 lit windows per frame), and a plain batch with ONE dense4 frame in it (a camera frame with a lit
        # window must not stall its launch).  Four frame sets per level (consecutive steps never share input).
        t_sw = time.perf_counter()
        levels = [("plain", 0, False), ("dense2", 12, False), ("dense4", 14, False), ("plain + one dense4 frame per batch", 0, True)]
        if args.density_sweep:
            levels = [("plain", 0, False), ("dense1", 11, False), ("dense2", 12, False), ("dense3", 13, False), ("dense4", 14, False),
                      ("plain + one dense4 frame per batch", 0, True)]
        barrier()
        pl_d = make_pipeline(ns, args.pixel_streams, args.sparse_streams, mc=4096)
        sweep, region = [], max(40, 2 * args.steps)
        for label, var, one in levels:
            sets_, _ = (frames_k[:4], None) if (var == 0 and not one) else frame_sets(4, W, H, var, one)
            cur["pl"], cur["sets"] = pl_d, sets_
            # warm-up in two goes with a drain between: the pipeline's policies (hot contexts, the dense frames' second launch) follow the
            # records that have COME BACK -- after the drain the second go runs under the level's own policy, so the timed regions start
            # in the level's steady state (round 4's sweep timed the policy's switch-over with the first steps of its one region)
            for _ in range(2 * ns):
                step()
            barrier()
            for _ in range(ns):
                step()
            n_sub0 = len(submit_max_each)
            rep_sw, _ = regions(region, 3)
            each_sw = [x / region for x in rep_sw]
            dsw = median(each_sw)
            info_d = pl_d.get_info()
            cnt_ = pl_d.context_of(info_d.submitted - 1).counts()
            st_ = cnt_["status"]
            sweep.append({"stream": label, "ms_per_step": round(dsw * 1e3, 4), "ms_per_step_each": [round(x * 1e3, 4) for x in each_sw],
                          "max_submit_host_ms": round(max(submit_max_lib_each[n_sub0:]) * 1e3, 4), "max_submit_host_ms_python_clock": round(max(submit_max_each[n_sub0:]) * 1e3, 4),
                          "host_blocking_calls": int(info_d.host_blocking_calls),
                          "frames_per_s": round(n / dsw, 1),
                          "contours_per_frame": round(float(cnt_["n_contours"].mean()), 1),
                          "points_per_frame": round(float(cnt_["n_points"].mean()), 1),
                          "frames_mid_tier": int(np.count_nonzero(st_ & 64)), "frames_slow_path": int(np.count_nonzero(st_ & 16)),
                          "frames_over_capacity": int(np.count_nonzero(st_ & 15))})
            del sets_
        barrier()
        cur["pl"], cur["sets"] = pl, frames_k
        pl_d.close()
        base_ = sweep[0]["ms_per_step"]
        for lv in sweep:
            lv["x_plain"] = round(lv["ms_per_step"] / base_, 3)
        out["density_sweep"] = {"steps_per_region": region, "levels": sweep, "seconds": round(time.perf_counter() - t_sw, 1),
                                "note": "three steady-state regions per level of the steps' own loop (%d batches in flight over %d sparse streams, 4 frame sets per "
                                        "level), the median; max_submit_host_ms = the longest single rmcv_pipeline_submit call of the level's regions by the library's own clock (rmcv_pipeline_info::max_submit_us); "
                                        "x_plain = against this sweep's own plain level; not the metric" % (info.depth, info.sparse_streams)}

    if extras and args.workload == "c3" and args.variant == 0 and not args.pose and not args.one_dense:
        # ---- BASELINE config 5 in short: 256 x 1920x1200 + icon rectification + SVM, the steps' own schedule over 4 frame sets
        t_c5 = time.perf_counter()
        W5, H5 = WORKLOADS["c5"]
        barrier()
        sets5, _ = frame_sets(4, W5, H5, 0)
        svm5 = synth.svm_weights()
        pl5 = make_pipeline(ns, args.pixel_streams, args.sparse_streams, w=W5, h=H5, mc=2048, with_svm=svm5)
        cur.update(pl=pl5, sets=sets5, w=W5, h=H5, stages=STAGE_ALL | STAGE_IDENTITY)
        for _ in range(2 * ns):
            step()
        rep5, _ = regions(20, 3)
        d5 = median(rep5) / 20
        last5 = pl5.get_info().submitted - 1
        a5, _o5 = pl5.collect(last5)
        ident5 = pl5.context_of(last5).identities()
        barrier()
        cur.update(pl=pl, sets=frames_k, w=W, h=H, stages=stages)
        out["c5"] = {"workload": "C5: batch=%d %dx%d BGR, full path + icon rectification + 7-class linear SVM (synthetic weights)" % (n, W5, H5),
                     "ms_per_step": round(d5 * 1e3, 4), "frames_per_s": round(n / d5, 1), "hbm_frac": round(n * 4 * W5 * H5 / d5 / 1e9 / HBM_PEAK_GBS, 4),
                     "batches_in_flight": pl5.info.depth, "regions": "3 x 20 steps (median)", "armours_last_batch": int(len(a5)),
                     "identities_last_batch": int(len(ident5)), "seconds": round(time.perf_counter() - t_c5, 1)}
        pl5.close()
        del sets5

    if extras:
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
        # the same chain from a C host (tools/frame_chain.c): what an unchanged C++ caller sees -- the figures above carry three ctypes
        # calls per frame.  Measured twice: FIRST THING in this run, before this process had touched the GPU (c_host: a host has the
        # GPU to itself, as the reference's executable does), and here, as a child of a process that holds a dozen queues on the same
        # GPU (c_host_beside_this_process: the runtime's own waits get slower then -- the library's do not depend on them any more)
        if c_host_first is not None:
            sf["c_host"] = c_host_first
        sf["c_host_beside_this_process"] = c_host_chain(W, H)
        sf["c_host_beside_this_process_runtime_copies_only"] = c_host_chain(W, H, 0)  # RMCV_OPT_FRAME_UPLOAD 0 + RMCV_OPT_IMAGE_EXPORT 0: round 4's chain
        sf["c_host_beside_this_process_own_paths_only"] = c_host_chain(W, H, 1)       # pinned staging + export kernel: nothing of the runtime's pageable copies
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
    barrier()                                       # rank 0 has the extras and the CPU baseline to itself: leave together
    pl.close()
    if abi_gather is not None:
        abi_gather.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
