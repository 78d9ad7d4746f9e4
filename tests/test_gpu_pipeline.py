"""rmcv_pipeline_* (include/rmcv_abi.h): the pipelined batch schedule behind the C-ABI -- the process loop of the reference
(executable/main.cpp:163-209) in batch form.  Streams of DISTINCT batches go through a depth-8 ring; every batch's armour list must
come back equal to the oracle's, in submission order, whatever else is in flight beside it."""
import ctypes as C
import json
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from rmcv_amd import (CAMP_BLUE, STAGE_ALL, STAGE_BINARY, STAGE_CONTOURS, STAGE_IDENTITY, STAGE_NO_IMAGE, LegacyParams, Pipeline, RmcvError,
                      default_params, synth)
from rmcv_amd import abi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_lists(oracle, frames):
    with ThreadPoolExecutor(16) as ex:
        return list(ex.map(lambda f: oracle.detect_frame(f)["armours"], frames))


def check_batch(oracle, frames, arm, offs):
    refs = oracle_lists(oracle, frames)
    assert len(offs) == len(frames) + 1
    for f, r in enumerate(refs):
        assert arm[offs[f]:offs[f + 1]].tobytes() == r.tobytes(), f
    assert offs[-1] == sum(len(r) for r in refs)


@pytest.mark.parametrize("one_dense,dense_streams", [(False, 0), (True, 0), (True, -1), (True, 1)],
                         ids=["plain", "one_dense_frame_per_batch", "one_dense_no_second_launch", "one_dense_one_dense_stream"])
def test_24_distinct_batches_through_a_depth_8_pipeline(oracle, one_dense, dense_streams):
    """24 distinct 256-frame batches (1280x1024), eight in flight: the list of every batch equals the oracle's, in submission order.
    With one frame per batch beyond findContours' LDS tables (a lit window + 2000 specks: the mid tier), which finishes long after
    its batch's other frames."""
    import torch
    dev = torch.device("cuda", 0)
    n, w, h, nb = 256, 1280, 1024, 24
    pl = Pipeline(device=0, depth=8, dense_streams=dense_streams, max_frames=n, max_width=w, max_height=h, max_contours=4096)
    assert (pl.info.depth, pl.info.pixel_streams, pl.info.sparse_streams, pl.info.sparse_waves, pl.info.pixel_groups) == (8, 2, 4, 4, 2)
    assert pl.info.dense_streams == {0: 4, -1: 0, 1: 1}[dense_streams]
    p = default_params()
    host, devf, got = [], [], {}
    for i in range(nb):
        fr = synth.batch(500000 + 7919 * i, n, w, h, CAMP_BLUE, i % 2, threads=16)
        if one_dense:
            fr[(37 * i) % n] = synth.frame(900000 + i, w, h, CAMP_BLUE, 14)
        host.append(fr)
        devf.append(torch.from_numpy(fr).to(dev))
        t = pl.submit(devf[i].data_ptr(), n, h, w, p, STAGE_ALL)
        assert t == i
        if i >= 8:                                                    # the slot of ticket i - 8 has just been handed to ticket i ...
            with pytest.raises(RmcvError):
                pl.collect(i - 8)
        if i >= 7:                                                    # ... so its predecessor is collected while seven others are in flight
            got[i - 7] = pl.collect(i - 7)
    pl.drain()
    for i in range(nb - 7, nb):
        got[i] = pl.collect(i)
    with pytest.raises(RmcvError):
        pl.collect(0)                                                  # long gone
    with pytest.raises(RmcvError):
        pl.collect(nb)                                                 # never issued
    for i in range(nb):
        arm, offs = got[i]
        check_batch(oracle, host[i], arm, offs)
    if one_dense:
        st = pl.context_of(nb - 1).counts()["status"]
        assert np.count_nonzero(st & abi.FRAME_MID_PATH) == 1 and not (st & 15).any()
    assert pl.get_info().submitted == nb
    # the dense frames get their own launch and stream from the second ring cycle on (the first cycle has nothing to go by), and only
    # where there are some
    assert pl.get_info().host_blocking_calls == 0                      # ... nor when a slot's finishing stream changes (the dense frames' second launch comes and goes)
    split = pl.get_info().dense_split
    assert (split >= nb - 2 * 8) if (one_dense and dense_streams >= 0) else split == 0
    # calm batches (no frame beyond the LDS tables in what came back) take turns at four contexts and run the wave-specialised pixel
    # kernel -- from the second ring cycle on; a stream with a dense frame in every batch never does
    hot = pl.get_info().hot_batches
    assert pl.get_info().hot_contexts == 4
    assert (hot >= nb - 2 * 8) if not one_dense else hot == 0
    pl.close()


def test_pipeline_stage_getters_sizes_and_modes(oracle):
    """smaller geometries, ragged batch sizes, the geometry changing in mid-stream, lists kept on the device (host_results = 2), a
    detection-only stage mask; per-stage getters on the slot's context"""
    import torch
    dev = torch.device("cuda", 0)
    pl = Pipeline(device=0, depth=3, pixel_streams=2, sparse_streams=2, host_results=2, max_frames=40, max_width=1920, max_height=1200)
    assert pl.info.host_results == 2 and pl.info.depth == 3
    p = default_params()
    plan = [(40, 640, 512, STAGE_ALL), (13, 1920, 1200, STAGE_ALL), (40, 1280, 1024, STAGE_ALL | STAGE_NO_IMAGE), (1, 1280, 720, STAGE_ALL),
            (40, 640, 512, STAGE_ALL), (7, 1280, 1024, STAGE_ALL), (40, 1920, 1200, STAGE_ALL)]
    keep = []
    for i, (n, w, h, st) in enumerate(plan):
        fr = synth.batch(31000 + 100 * i, n, w, h, CAMP_BLUE, i % 2, threads=16)
        d = torch.from_numpy(fr).to(dev)
        keep.append((fr, d))
        t = pl.submit(d.data_ptr(), n, h, w, p, st)
        arm, offs = pl.collect(t)
        check_batch(oracle, fr, arm, offs)
        c = pl.context_of(t)
        f = n // 2
        r = oracle.detect_frame(fr[f])
        pts, co = c.contours(f)
        assert np.array_equal(co, r["offs"]) and np.array_equal(pts, r["pts"])
        assert c.blobs(f)[0].tobytes() == r["blobs"].tobytes()
        if not (st & STAGE_NO_IMAGE):
            assert np.array_equal(c.binary(f), r["binary"])
    with pytest.raises(RmcvError):
        pl.submit(keep[0][1].data_ptr(), 40, 512, 640, p, STAGE_ALL & ~STAGE_BINARY)     # a pipelined batch starts at the pixel kernel
    with pytest.raises(RmcvError):
        pl.submit(keep[0][1].data_ptr(), 41, 512, 640, p, STAGE_ALL)                      # more frames than the ring's contexts hold
    with pytest.raises(RmcvError):
        pl.submit(keep[0][1].data_ptr(), 40, 512, 640, p, STAGE_ALL | STAGE_IDENTITY)    # no SVM loaded: refused BEFORE anything is enqueued ...
    t = pl.submit(keep[0][1].data_ptr(), 40, 512, 640, p, STAGE_ALL)                      # ... so the slot is as it was
    arm, offs = pl.collect(t)
    check_batch(oracle, keep[0][0], arm, offs)
    t = pl.submit(keep[0][1].data_ptr(), 40, 512, 640, p, STAGE_BINARY | STAGE_CONTOURS)  # a partial path is fine (its list is the previous run's)
    pl.wait(t)
    pts, co = pl.context_of(t).contours(3)
    r = oracle.detect_frame(keep[0][0][3])
    assert np.array_equal(co, r["offs"]) and np.array_equal(pts, r["pts"])
    pl.close()


def test_pipeline_capacity_errors_name_the_need():
    import torch
    dev = torch.device("cuda", 0)
    n, w, h = 16, 1280, 1024
    fr = synth.batch(4200, n, w, h, CAMP_BLUE, 0, threads=16)
    d = torch.from_numpy(fr).to(dev)
    pl = Pipeline(device=0, depth=2, armour_cap=3, max_frames=n, max_width=w, max_height=h)
    t = pl.submit(d.data_ptr(), n, h, w, default_params(), STAGE_ALL)
    out = np.empty(64, abi.ARMOUR)
    offs = np.empty(n + 1, np.int32)
    tot = C.c_int32(0)
    rc = abi.lib().rmcv_pipeline_collect(pl._h, t, abi.ptr(out), 64, abi.ptr(offs), C.addressof(tot))
    assert rc == abi.ERR_CAPACITY and tot.value > 3 and offs[n] == tot.value         # the ring's armour_cap is too small: says how many there are
    pl.close()
    pl = Pipeline(device=0, depth=2, max_frames=n, max_width=w, max_height=h, max_contours=4)
    t = pl.submit(d.data_ptr(), n, h, w, default_params(), STAGE_ALL)
    with pytest.raises(RmcvError) as e:
        pl.collect(t)
    assert e.value.code == abi.ERR_CAPACITY                                           # a frame exceeded a context limit: the status word says so
    assert (pl.context_of(t).counts()["status"] & abi.FRAME_OVF_CONTOURS).any()
    pl.close()


def test_pipeline_hook_sees_every_record_in_order():
    """the hook rides behind every batch's compaction, on the stream the record is produced on; what it enqueues there sees the
    finished record.  Here: a device-to-device copy of the record into a log (torch, on the hook's stream)."""
    import torch
    from rmcv_amd import dist as rdist
    dev = torch.device("cuda", 0)
    n, w, h, depth, nb = 32, 1280, 1024, 4, 10
    pl = Pipeline(device=0, depth=depth, max_frames=n, max_width=w, max_height=h)
    rb = pl.info.record_bytes
    log = torch.zeros((nb, rb), dtype=torch.uint8, device=dev)
    seen = []

    def hook(ticket, d_record, nbytes, stream):
        assert nbytes == rb
        seen.append(ticket)
        with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
            log[ticket].copy_(rdist.tensor_at(d_record, nbytes, dev), non_blocking=True)
        return None
    pl.set_hook(hook)
    sets = [torch.from_numpy(synth.batch(8800 + 50 * i, n, w, h, CAMP_BLUE, 0, threads=16)).to(dev) for i in range(nb)]
    for i in range(nb):
        pl.submit(sets[i].data_ptr(), n, h, w, default_params(), STAGE_ALL)
    pl.drain()
    torch.cuda.synchronize()
    assert seen == list(range(nb))
    head = pl.info.armours_offset
    for i in range(nb - depth, nb):
        arm, offs = pl.collect(i)
        rec = log[i].cpu().numpy()
        assert rec[:(n + 1) * 4].view(np.int32).tolist() == offs.tolist()
        assert rec[head:head + len(arm) * 88].tobytes() == arm.tobytes()
    for i in range(nb - depth):                                       # the log keeps what the ring has long overwritten
        assert log[i].cpu().numpy()[:(n + 1) * 4].view(np.int32)[n] > 0
    pl.set_hook(None)
    pl.close()


def test_pipeline_identity_and_legacy_stages(oracle):
    """RMCV_STAGE_IDENTITY (BASELINE config 5) and the legacy blob stage through the ring equal the plain batch entry points"""
    import torch
    from rmcv_amd import Context
    dev = torch.device("cuda", 0)
    n, w, h = 12, 1920, 1200
    fr = synth.batch(61000, n, w, h, CAMP_BLUE, 0, threads=16)
    d = torch.from_numpy(fr).to(dev)
    svm = synth.svm_weights()
    pl = Pipeline(device=0, depth=2, max_frames=n, max_width=w, max_height=h)
    for c in pl.contexts:
        c.svm_load(*svm)
    ref = Context(device=0, max_frames=n, max_width=w, max_height=h)
    ref.svm_load(*svm)
    ref.upload(fr)
    ref.run(default_params(), STAGE_ALL | STAGE_IDENTITY)
    ref.sync()
    t = pl.submit(d.data_ptr(), n, h, w, default_params(), STAGE_ALL | STAGE_IDENTITY)
    arm, offs = pl.collect(t)
    ra, ro = ref.armours()
    assert arm.tobytes() == ra.tobytes() and offs.tolist() == ro.tolist() and len(arm) > 0
    assert pl.context_of(t).identities().tolist() == ref.identities().tolist()
    lp = LegacyParams(1.5, 80, 70, 10, 99999, 0)
    ref.run_legacy(lp, default_params(), STAGE_ALL)
    ref.sync()
    t = pl.submit(d.data_ptr(), n, h, w, default_params(), STAGE_ALL, legacy=lp)
    arm, offs = pl.collect(t)
    ra, ro = ref.armours()
    assert arm.tobytes() == ra.tobytes() and offs.tolist() == ro.tolist()
    ref.close()
    pl.close()


def test_c_host_drives_the_same_pipeline(oracle):
    """tools/pipeline_bench.c: the three calls from a C program (no Python, no torch in that process) -- a small run, its armour
    counts against the oracle's for the same synthetic frames"""
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        pytest.skip("no C compiler on this box")
    exe = os.path.join(ROOT, "gpurun_out", "pipeline_bench_test")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir = os.path.join(ROOT, "rmcv_amd", "lib")
    subprocess.run([cc, "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "pipeline_bench.c"), "-o", exe, "-L", libdir,
                    "-lrmcv_hip", "-Wl,-rpath," + libdir, "-lpthread"], check=True, timeout=120)
    n, depth, sets, steps = 16, 4, 4, 6
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}   # the host's own rmcv_hw_queues_hint() applies
    cp = subprocess.run([exe, "--frames", str(n), "--depth", str(depth), "--sets", str(sets), "--steps", str(steps), "--repeats", "2", "--warmup", "2",
                         "--warmup-seconds", "0.05"], capture_output=True, text=True, timeout=600, env=env)
    assert cp.returncode == 0, cp.stderr
    out = json.loads(cp.stdout.strip().splitlines()[-1])
    assert out["steps"] == steps and out["depth"] == depth and out["value"] > 0
    assert out["gpu_max_hw_queues"] == 12                                      # set by the host's rmcv_hw_queues_hint() before its first HIP call
    per_set = [sum(len(a) for a in oracle_lists(oracle, synth.batch(k * 1000003, n, 1280, 1024, CAMP_BLUE, 0, threads=16))) for k in range(sets)]
    assert out["armours_set0"] == per_set[0]
    assert out["armours_last_%d_batches" % depth] == sum(per_set)               # the last four steps cover every set once (sets == depth)


def test_pipeline_gathers_single_rank(oracle):
    """BASELINE config 4's plumbing on one GPU: the built-in hook (rmcv_pipeline_set_gather: rmcv_gather on an rmcv_comm, here a group
    of one) and the torch.distributed hook (rmcv_amd.dist.TorchGatherHook on a one-rank nccl group) both deliver every batch's record,
    ticket by ticket, while the ring keeps running"""
    import socket

    import torch
    import torch.distributed as dist
    from rmcv_amd import dist as rdist
    dev = torch.device("cuda", 0)
    n, w, h, depth, nb = 24, 1280, 1024, 3, 9
    sets = [synth.batch(120000 + 300 * i, n, w, h, CAMP_BLUE, i % 2, threads=16) for i in range(nb)]
    dsets = [torch.from_numpy(s).to(dev) for s in sets]
    refs = [oracle_lists(oracle, s) for s in sets]

    def check(i, recs, cap):
        arm, offs = rdist.unpack_records(recs, n, cap)
        assert offs.tolist() == np.cumsum([0] + [len(r) for r in refs[i]]).tolist(), i
        assert arm.tobytes() == np.concatenate(refs[i]).tobytes(), i
    # ---- rmcv_gather through rmcv_pipeline_set_gather
    L = abi.lib()
    idb = (C.c_uint8 * abi.COMM_ID_BYTES)()
    assert L.rmcv_comm_unique_id(idb) == 0
    hc = C.c_void_p()
    assert L.rmcv_comm_create(idb, 1, 0, 0, C.byref(hc)) == 0
    pl = Pipeline(device=0, depth=depth, max_frames=n, max_width=w, max_height=h)
    pl.set_gather(hc, 0)
    assert pl.get_info().hw_queues_wanted == 1 + 2 + 3 + min(4, depth) + 1
    for i in range(nb):
        t = pl.submit(dsets[i].data_ptr(), n, h, w, default_params(), STAGE_ALL)
        if i >= depth - 1:
            j = i - (depth - 1)
            pl.wait(j)
            d, nbytes = pl.gathered(j)
            assert nbytes == pl.info.record_bytes
            check(j, [rdist.tensor_at(d, nbytes, dev)], pl.info.armour_cap)
    pl.close()
    L.rmcv_comm_destroy.restype = None
    L.rmcv_comm_destroy.argtypes = [C.c_void_p]
    L.rmcv_comm_destroy(hc)
    # ---- torch.distributed through the hook
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        pl = Pipeline(device=0, depth=depth, max_frames=n, max_width=w, max_height=h)
        hook = rdist.TorchGatherHook(pl.info.record_bytes, depth, dev)
        pl.set_hook(hook)
        for i in range(nb):
            pl.submit(dsets[i].data_ptr(), n, h, w, default_params(), STAGE_ALL)
            if i >= depth - 1:
                j = i - (depth - 1)
                hook.wait(j)
                check(j, hook.records(j), pl.info.armour_cap)
        pl.drain()
        hook.wait_all()
        pl.set_hook(None)
        pl.close()
    finally:
        dist.destroy_process_group()


def test_calm_and_dense_batches_in_turn(oracle):
    """a stream that changes character -- plain batches, then batches full of dense frames, then plain again: the pipeline moves between
    four contexts in rotation (+ the wave-specialised pixel kernel) and one context per slot; every list equals the oracle's, and a
    waited-for ticket's context holds THAT batch's stages"""
    import torch
    dev = torch.device("cuda", 0)
    n, w, h = 64, 1280, 1024
    pl = Pipeline(device=0, max_frames=n, max_width=w, max_height=h, max_contours=4096, hot_contexts=4)
    assert pl.info.depth == 8 and pl.info.hot_contexts == 4
    p = default_params()
    kinds = [0] * 12 + [14] * 5 + [1] * 14 + [14] * 3 + [0] * 10
    ws0 = abi.lib().rmcv_pixel_ws_launches()
    host, devf, got = [], [], {}
    for i, kind in enumerate(kinds):
        fr = synth.batch(770000 + 31 * i, n, w, h, CAMP_BLUE, kind if kind != 14 else 0, threads=16)
        if kind == 14:
            fr[::5] = synth.batch(880000 + i, len(fr[::5]), w, h, CAMP_BLUE, 14, threads=16)
        host.append(fr)
        devf.append(torch.from_numpy(fr).to(dev))
        t = pl.submit(devf[i].data_ptr(), n, h, w, p, STAGE_ALL)
        if i % 7 == 3:                                                # now and then: wait and look at the batch's own context
            pl.wait(t)
            c = pl.context_of(t)
            r = oracle.detect_frame(fr[n // 2])
            pts, co = c.contours(n // 2)
            assert np.array_equal(co, r["offs"]) and np.array_equal(pts, r["pts"])
            assert np.array_equal(c.binary(n // 2), r["binary"])
        if i >= 7:
            got[i - 7] = pl.collect(i - 7)
    pl.drain()
    for i in range(len(kinds) - 7, len(kinds)):
        got[i] = pl.collect(i)
    for i in range(len(kinds)):
        check_batch(oracle, host[i], *got[i])
    info = pl.get_info()
    assert 10 <= info.hot_batches <= len(kinds) - 1 - 6               # most ran hot; the first one and the dense ones (but for the first of a run) did not
    assert abi.lib().rmcv_pixel_ws_launches() - ws0 == info.hot_batches
    # rmcv_pipeline_submit never blocks the host: no allocation, no synchronisation, no blocking copy -- not at the first batch, not when
    # the stream turns dense (the slots' finishing streams change), not when it turns calm again
    assert info.host_blocking_calls == 0
    pl.close()


def test_geometry_changes_with_batches_in_flight(oracle):
    """frame sizes and batch sizes change from one submit to the next while eight batches are in flight and the calm ones share four
    contexts: a context re-zeroes its planes and rewrites its frame order only when its own last batch is through"""
    import torch
    dev = torch.device("cuda", 0)
    pl = Pipeline(device=0, max_frames=96, max_width=1920, max_height=1200)
    assert pl.info.depth == 8 and pl.info.hot_contexts == 7            # derived: 96 x 1920x1200 bit planes = 29.5 MB per batch, 200 MiB of cache -> the whole ring but one
    p = default_params()
    geoms = [(96, 1280, 1024), (40, 1920, 1200), (96, 640, 512), (17, 1280, 720), (96, 1280, 1024), (64, 1920, 1080)]
    host, devf, got = [], [], {}
    for i in range(30):
        n, w, h = geoms[(i * 7) % len(geoms)] if i % 3 else geoms[i % len(geoms)]
        fr = synth.batch(660000 + 13 * i, n, w, h, CAMP_BLUE, i % 2, threads=16)
        host.append(fr)
        devf.append(torch.from_numpy(fr).to(dev))
    for i in range(30):
        n, h, w, _ = host[i].shape
        pl.submit(devf[i].data_ptr(), n, h, w, p, STAGE_ALL)
        if i >= 7:
            got[i - 7] = pl.collect(i - 7)
    pl.drain()
    for i in range(23, 30):
        got[i] = pl.collect(i)
    for i in range(30):
        check_batch(oracle, host[i], *got[i])
    assert pl.get_info().hot_batches > 0
    assert pl.get_info().host_blocking_calls == 0                      # a change of geometry is enqueued work (planes zeroed, frame order on the device)
    pl.close()


def test_hot_contexts_config_and_switch():
    """rmcv_pipeline_config::hot_contexts: the default, off, out of range, what it needs; rmcv_pipeline_set_hot_contexts at run time"""
    import torch
    dev = torch.device("cuda", 0)
    n, w, h = 32, 640, 512
    pl = Pipeline(device=0, max_frames=n, max_width=w, max_height=h)
    assert pl.info.hot_contexts == 7                                  # derived: small planes, the whole ring but one
    with pytest.raises(RmcvError):
        pl.set_hot_contexts(2)                                        # fewer than 3 stall even sparse batches
    with pytest.raises(RmcvError):
        pl.set_hot_contexts(8)                                        # the ring has 8 contexts: 3 .. 7
    pl.set_hot_contexts(5)
    assert pl.get_info().hot_contexts == 5
    pl.set_hot_contexts(0)
    assert pl.get_info().hot_contexts == 0
    fr = synth.batch(99, n, w, h, CAMP_BLUE, 0, threads=16)
    d = torch.from_numpy(fr).to(dev)
    p = default_params()
    for i in range(12):
        pl.submit(d.data_ptr(), n, h, w, p, STAGE_ALL)
        pl.wait(i)
    assert pl.get_info().hot_batches == 0                             # off: a context per slot
    pl.set_hot_contexts(4)
    for i in range(12, 24):
        pl.submit(d.data_ptr(), n, h, w, p, STAGE_ALL)
        pl.wait(i)
    assert pl.get_info().hot_batches == 12
    pl.close()
    for kw, want in ((dict(hot_contexts=-1), 0), (dict(hot_contexts=6), 6), (dict(hot_contexts=9), 0), (dict(depth=3), 0), (dict(host_results=2), 0),
                     (dict(sparse_waves=8), 0)):
        q = Pipeline(device=0, max_frames=n, max_width=w, max_height=h, **kw)
        assert q.info.hot_contexts == want, kw
        q.close()


def test_the_newest_batch_is_finished_by_the_call_that_waits_for_it(oracle):
    """the back half of the newest batch is enqueued by the next call: by the next submit as ever, by a wait / collect of that very
    ticket or a drain with the latency kernel (rmcv_pipeline_info::latency_batches); results are the same either way"""
    import torch
    dev = torch.device("cuda", 0)
    n, w, h = 48, 1280, 1024
    pl = Pipeline(device=0, max_frames=n, max_width=w, max_height=h)
    p = default_params()
    fr = [synth.batch(5150 + i, n, w, h, CAMP_BLUE, i % 2, threads=16) for i in range(6)]
    d = [torch.from_numpy(f).to(dev) for f in fr]
    t0 = pl.submit(d[0].data_ptr(), n, h, w, p, STAGE_ALL)
    t1 = pl.submit(d[1].data_ptr(), n, h, w, p, STAGE_ALL)              # finishes t0 the ordinary way
    assert pl.get_info().latency_batches == 0
    check_batch(oracle, fr[0], *pl.collect(t0))                        # an older ticket: t1's back half goes out the ordinary way too
    assert pl.get_info().latency_batches == 0
    check_batch(oracle, fr[1], *pl.collect(t1))
    t2 = pl.submit(d[2].data_ptr(), n, h, w, p, STAGE_ALL)
    check_batch(oracle, fr[2], *pl.collect(t2))                        # the newest ticket itself: the latency kernel
    assert pl.get_info().latency_batches == 1
    t3 = pl.submit(d[3].data_ptr(), n, h, w, p, STAGE_ALL)
    pl.drain()
    assert pl.get_info().latency_batches == 2
    check_batch(oracle, fr[3], *pl.collect(t3))
    t4 = pl.submit(d[4].data_ptr(), n, h, w, p, STAGE_ALL)
    c = pl.context_of(t4)                                              # names the newest ticket: finishes it (latency kernel), does not wait
    pl.wait(t4)
    r = oracle.detect_frame(fr[4][7])
    pts, co = c.contours(7)
    assert np.array_equal(co, r["offs"]) and np.array_equal(pts, r["pts"])
    assert pl.get_info().latency_batches == 3
    t5 = pl.submit(d[5].data_ptr(), n, h, w, p, STAGE_ALL)
    pl.close()                                                         # a pipeline destroyed with a batch in hand finishes it first


def test_dense_mode_follows_the_stream(oracle):
    """a stream that turns heavy -- 400 specks per frame (dense2: frames the LDS tier still holds), then 2 000 (dense4: every frame beyond
    it) -- and plain again: while the records say so the batches run in dense mode (the lean build of the sparse kernel: every frame on the
    mid tier, 61 KB of LDS instead of 80) and leave it when the stream calms down; every list equals
    the oracle's whichever kernel produced it; submit never blocks"""
    import torch
    dev = torch.device("cuda", 0)
    n, w, h = 48, 1280, 1024
    pl = Pipeline(device=0, max_frames=n, max_width=w, max_height=h, max_contours=4096)
    p = default_params()
    kinds = [0] * 10 + [12] * 12 + [14] * 10 + [0] * 14
    host, devf, got = [], [], {}
    heavy_at = []
    for i, kind in enumerate(kinds):
        fr = synth.batch(990000 + 37 * i, n, w, h, CAMP_BLUE, kind, threads=16)
        host.append(fr)
        devf.append(torch.from_numpy(fr).to(dev))
        pl.submit(devf[i].data_ptr(), n, h, w, p, STAGE_ALL)
        heavy_at.append(pl.get_info().heavy_batches)
        if i >= 7:
            got[i - 7] = pl.collect(i - 7)
    pl.drain()
    for i in range(len(kinds) - 7, len(kinds)):
        got[i] = pl.collect(i)
    for i in range(len(kinds)):
        check_batch(oracle, host[i], *got[i])
    info = pl.get_info()
    is_heavy = np.diff([0] + heavy_at) > 0
    assert not is_heavy[:10].any()                                     # the plain stream: never
    assert is_heavy[12:32].sum() >= 8                                  # the dense part: once its first records are back
    assert not is_heavy[-4:].any() and info.hot_batches > 0            # plain again: out of dense mode, back into the hot contexts
    assert info.host_blocking_calls == 0
    st = pl.context_of(len(kinds) - 1).counts()["status"]
    assert not (st & abi.FRAME_MID_PATH).any()                         # the last plain batch ran the standard kernel
    pl.close()
