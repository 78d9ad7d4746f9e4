"""A drop-in that cannot hang its caller.  The reference's process loop is real-time and latest-wins
(/root/reference/executable/main.cpp:157, 169, 197): no entry point parks its caller in the HIP runtime without a bound.  Every wait
polls with a deadline (RMCV_OPT_WAIT_TIMEOUT_MS, rmcv_pipeline_set_wait_timeout); one that runs out returns RMCV_ERR_TIMEOUT and names
the kernel or copy enqueued last.  The stand-in for a kernel that does not finish in time is RMCV_OPT_TEST_DELAY_US: one sleeping
wavefront in front of the call's own kernels."""
import ctypes as C
import time

import numpy as np
import pytest

from rmcv_amd import (CAMP_BLUE, OPT_FRAME_UPLOAD, OPT_IMAGE_EXPORT, OPT_TEST_DELAY_US, OPT_TEST_SLOW_US, OPT_WAIT_TIMEOUT_MS, STAGE_ALL, Context, Pipeline, RmcvError, default_params, synth)
from rmcv_amd import abi

pytestmark = pytest.mark.gpu


def chain(ctx, img):
    pts, offs, binary = ctx.extract_color_csr(img)
    blobs, src, neg = ctx.filter_lightblobs(pts, offs)
    return pts, offs, binary, blobs, ctx.filter_armours(blobs)


@pytest.mark.parametrize("image_export", [0, 1])
def test_per_frame_call_gives_up_at_the_deadline_and_names_the_work(oracle, image_export):
    c = Context(device=0, max_frames=1, max_width=1280, max_height=1024)
    c.set_option(OPT_IMAGE_EXPORT, image_export)
    img = synth.frame(3, 1280, 1024, CAMP_BLUE, 0)
    ref = oracle.detect_frame(img)
    pts, offs, binary, blobs, arm = chain(c, img)                     # an ordinary frame first
    assert arm.tobytes() == ref["armours"].tobytes()
    c.set_option(OPT_WAIT_TIMEOUT_MS, 40)
    c.set_option(OPT_TEST_DELAY_US, 400000)                           # the frame's kernels sit behind 0.4 s of sleep
    # (buffers that outlive the failed call: after RMCV_ERR_TIMEOUT the work is still in flight and they stay borrowed -- see the header)
    L = abi.lib()
    keep_b = np.empty((1024, 1280), np.uint8)
    keep_p, keep_o = np.empty(c.limits.max_points, abi.POINT), np.empty(c.limits.max_contours + 1, np.int32)
    nc, npt = C.c_int32(0), C.c_int32(0)

    def call():
        return L.rmcv_extract_color(c._h, abi.ptr(img), 1280, 1024, 3 * 1280, CAMP_BLUE, 80, 2, abi.ptr(keep_b), abi.ptr(keep_p), len(keep_p),
                                    abi.ptr(keep_o), len(keep_o) - 1, C.byref(nc), C.byref(npt))
    t0 = time.perf_counter()
    rc = call()
    dt = time.perf_counter() - t0
    assert rc == abi.ERR_TIMEOUT
    assert 0.03 < dt < 0.3, dt                                        # gave up at the deadline, not at the kernel's end
    msg = L.rmcv_last_error(c._h).decode()
    assert "40 ms" in msg and "k_binary" in msg, msg                 # what did not finish
    # the work is still in flight: the next call first waits for it (same deadline) -- and fails the same way while it lasts
    assert call() == abi.ERR_TIMEOUT
    time.sleep(0.5)
    c.set_option(OPT_WAIT_TIMEOUT_MS, 5000)
    pts, offs, binary, blobs, arm = chain(c, img)                     # through: the context is as good as new
    assert np.array_equal(binary, ref["binary"]) and np.array_equal(pts, ref["pts"]) and arm.tobytes() == ref["armours"].tobytes()
    c.close()


def test_pipeline_wait_gives_up_at_the_deadline(oracle):
    import torch
    dev = torch.device("cuda", 0)
    n, w, h = 16, 1280, 1024
    pl = Pipeline(device=0, depth=4, max_frames=n, max_width=w, max_height=h)
    assert pl.info.wait_timeout_ms == 5000
    fr = synth.batch(424242, n, w, h, CAMP_BLUE, 0, threads=16)
    d = torch.from_numpy(fr).to(dev)
    p = default_params()
    t = pl.submit(d.data_ptr(), n, h, w, p, STAGE_ALL)
    arm0, offs0 = pl.collect(t)
    pl.set_wait_timeout(30)
    assert pl.get_info().wait_timeout_ms == 30
    for k in range(pl.depth):                                          # whichever context the next batch takes: its pixel launch sits behind 0.3 s
        pl.contexts[k].set_option(OPT_TEST_DELAY_US, 300000)
    t = pl.submit(d.data_ptr(), n, h, w, p, STAGE_ALL)
    t0 = time.perf_counter()
    with pytest.raises(RmcvError) as e:
        pl.collect(t)
    assert e.value.code == abi.ERR_TIMEOUT and time.perf_counter() - t0 < 0.25
    assert "30 ms" in str(e.value) and "enqueued last" in str(e.value)
    with pytest.raises(RmcvError) as e:
        pl.drain()
    assert e.value.code == abi.ERR_TIMEOUT
    pl.set_wait_timeout(5000)
    arm, offs = pl.collect(t)                                          # waiting again is allowed; the batch comes through unharmed
    assert arm.tobytes() == arm0.tobytes() and offs.tolist() == offs0.tolist()
    for k in range(pl.depth):
        pl.contexts[k].set_option(OPT_TEST_DELAY_US, 0)
    pl.drain()
    assert pl.get_info().host_blocking_calls == 0
    pl.close()


def test_small_bursts_are_not_held_back():
    """the hold-back of a burst's second pixel launch exists for launches of the wave-specialised kernel that fill every CU for > 100 us;
    two 16-frame batches (a 15 us pixel kernel) are never delayed, 256-frame bursts are"""
    import torch
    dev = torch.device("cuda", 0)
    w, h = 1280, 1024
    p = default_params()
    for n, want_held in ((16, False), (256, True)):
        pl = Pipeline(device=0, max_frames=n, max_width=w, max_height=h)
        fr = synth.batch(99000, n, w, h, CAMP_BLUE, 0, threads=16)
        d = torch.from_numpy(fr).to(dev)
        for burst in range(6):
            a = pl.submit(d.data_ptr(), n, h, w, p, STAGE_ALL)
            b = pl.submit(d.data_ptr(), n, h, w, p, STAGE_ALL)
            pl.collect(a)
            pl.collect(b)
            pl.drain()
        info = pl.get_info()
        assert info.hot_batches > 0
        assert (info.held_back > 0) == want_held, (n, info.held_back)
        pl.close()


def test_the_chain_leaves_the_runtimes_copies_while_they_are_slow(oracle):
    """RMCV_OPT_FRAME_UPLOAD 3 / RMCV_OPT_IMAGE_EXPORT 2 (the defaults): the runtime's pageable copies while they are fast; three slow
    frames in a row (here: RMCV_OPT_TEST_SLOW_US added to what the library measures) and the chain moves to the pinned staging buffer
    and the export kernel, and back when they have recovered; the results never change"""
    L = abi.lib()
    c = Context(device=0, max_frames=1, max_width=1280, max_height=1024)
    img = synth.frame(5, 1280, 1024, CAMP_BLUE, 1)
    ref = oracle.detect_frame(img)

    def paths():
        us = (C.c_double * 9)()
        assert L.rmcv_ctx_frame_timing(c._h, us, 9) == 0
        return int(us[7]), int(us[8])

    def one():
        pts, offs, binary, blobs, arm = chain(c, img)
        assert np.array_equal(binary, ref["binary"]) and np.array_equal(pts, ref["pts"]) and arm.tobytes() == ref["armours"].tobytes()
        return paths()
    one()
    c.set_option(OPT_FRAME_UPLOAD, 3)                                   # (the defaults, set again: the counters start over whatever the first frame met)
    c.set_option(OPT_IMAGE_EXPORT, 2)
    c.set_option(OPT_TEST_SLOW_US, 400)
    seen = [one() for _ in range(5)]
    assert seen[0] == (0, 0) and seen[3:] == [(1, 1)] * 2, seen         # the runtime's copies first; three slow frames, then the library's own paths
    c.set_option(OPT_TEST_SLOW_US, 0)
    later = [one() for _ in range(520)]
    assert later[0] == (1, 1) and (0, 0) in later[500:], later[500:]    # 512 frames on the library's paths, then the runtime's copies get another try
    # (they stay unless they ARE slow on this box at this moment -- the regime the switch exists for; the test does not depend on which)
    c.set_option(OPT_FRAME_UPLOAD, 0)
    c.set_option(OPT_IMAGE_EXPORT, 0)
    c.set_option(OPT_TEST_SLOW_US, 400)
    assert [one() for _ in range(5)][-1] == (0, 0)                      # pinned to the runtime's paths: no switch
    c.close()
