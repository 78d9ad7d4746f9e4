"""Frame-level hand-over: the per-frame sparse kernel of a batch runs BESIDE the batch's own pixel kernel and takes each frame
when k_binary has written its last strip (per-frame progress words under a launch label, release/acquire at agent scope).
Off by default (RMCV_OPT_HANDOVER: it measured equal on the pipelined bench and slower for a lone batch); these tests switch it on
and drive both forms -- a full run handed one stream, which the library forks onto the context's side stream, and the explicit form
bench.py --handover uses: pixel kernel on one stream, sparse stages with RMCV_STAGE_HANDOVER on another, several batches in flight."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from rmcv_amd import (CAMP_BLUE, OPT_HANDOVER, OPT_PIXEL_GROUPS, OPT_SPARSE_WAVES, STAGE_ALL, STAGE_BINARY, STAGE_HANDOVER, Context,
                      RmcvError, default_params, synth)

pytestmark = pytest.mark.gpu


def snapshot(c, n):
    arm, offs = c.armours()
    out = []
    for f in range(n):
        pts, co = c.contours(f)
        out.append((c.binary(f).tobytes(), pts.tobytes(), co.tobytes(), c.blobs(f)[0].tobytes(), arm[offs[f]:offs[f + 1]].tobytes()))
    return out


@pytest.mark.parametrize("w,h,n", [(1280, 1024, 256), (1920, 1200, 6), (1920, 1200, 37), (640, 512, 19)])
def test_pipelined_handover_equals_the_oracle(oracle, w, h, n):
    """bench.py's schedule in small: three contexts in flight, pixel kernels alternating over two streams, the sparse stages on a
    third with RMCV_STAGE_HANDOVER (no wait for the pixel kernel as a whole); twelve steps; every stage of every frame of every
    context against the oracle afterwards.  (6 and 37 frames of 1200 rows: strip counts that are no multiple of the 8 XCD queues,
    frames whose strips straddle two XCDs -- the planes then cross L2s.)"""
    import torch
    dev = torch.device("cuda", 0)
    nctx = 3
    ctxs, frames = [], []
    for k in range(nctx):
        c = Context(device=0, max_frames=n, max_width=w, max_height=h)
        c.set_option(OPT_HANDOVER, 1)
        c.set_option(OPT_SPARSE_WAVES, 4)
        c.set_option(OPT_PIXEL_GROUPS, 2)
        fr = synth.batch(4000 + 1000 * k, n, w, h, CAMP_BLUE, k % 2, threads=16)
        c.upload(fr)
        ctxs.append(c)
        frames.append(fr)
    sA = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    sB = torch.cuda.Stream(device=dev, priority=-1)
    done = [torch.cuda.Event() for _ in range(nctx)]
    p = default_params()
    for step in range(12):
        k = step % nctx
        a = sA[step % 2]
        if step >= nctx:
            a.wait_event(done[k])
        ctxs[k].run(p, STAGE_BINARY, a.cuda_stream)
        ctxs[k].run(p, (STAGE_ALL & ~STAGE_BINARY) | STAGE_HANDOVER, sB.cuda_stream)
        done[k].record(sB)
    torch.cuda.synchronize()
    for k in range(nctx):
        assert not ctxs[k].counts()["status"].any()
        with ThreadPoolExecutor(16) as ex:
            refs = list(ex.map(lambda f: oracle.detect_frame(frames[k][f]), range(n)))
        arm, offs = ctxs[k].armours()
        for f in range(n):
            r = refs[f]
            assert np.array_equal(ctxs[k].binary(f), r["binary"]), (k, f)
            pts, co = ctxs[k].contours(f)
            assert np.array_equal(co, r["offs"]) and np.array_equal(pts, r["pts"]), (k, f)
            assert ctxs[k].blobs(f)[0].tobytes() == r["blobs"].tobytes(), (k, f)
            assert arm[offs[f]:offs[f + 1]].tobytes() == r["armours"].tobytes(), (k, f)
        ctxs[k].close()


def test_handover_off_and_on_agree():
    """RMCV_OPT_HANDOVER = 0 (the sparse kernel starts when the whole pixel kernel is through) and the default give identical
    buffers, for the one-stream form (forked by the library) and for repeated runs of one context"""
    n = 64
    fr = synth.batch(77000, n, 1280, 1024, CAMP_BLUE, 1, threads=16)
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    c.upload(fr)
    c.set_option(OPT_HANDOVER, 0)
    c.run(default_params(), STAGE_ALL)
    c.sync()
    ref = snapshot(c, n)
    c.set_option(OPT_HANDOVER, 1)
    for _ in range(3):                      # back to back: the progress words carry the label of the launch, no reset in between
        c.run(default_params(), STAGE_ALL)
    c.sync()
    assert snapshot(c, n) == ref
    c.close()


def test_handover_flag_needs_its_pixel_kernel():
    c = Context(device=0, max_frames=4, max_width=640, max_height=512)
    c.set_option(OPT_HANDOVER, 1)
    c.upload(synth.batch(1, 4, 640, 512))
    with pytest.raises(RmcvError):
        c.run(default_params(), (STAGE_ALL & ~STAGE_BINARY) | STAGE_HANDOVER)       # no pixel kernel was ever enqueued on this context
    c.run(default_params(), STAGE_BINARY)
    with pytest.raises(RmcvError):
        c.run(default_params(), STAGE_ALL | STAGE_HANDOVER)                         # the flag goes with a run WITHOUT the binary stage
    c.run(default_params(), (STAGE_ALL & ~STAGE_BINARY) | STAGE_HANDOVER)
    c.sync()
    assert not c.counts()["status"].any()
    c.close()


@pytest.mark.parametrize("w,h,n", [(1920, 1200, 3), (1280, 720, 5), (1280, 1000, 4), (1280, 720, 1)])
def test_handover_small_batches_whose_last_strip_is_partial(oracle, w, h, n):
    """a launch with fewer strips than half the CUs hands every strip out as four 8-row pieces; with h % 32 in
    1..23 a piece of the last strip lies below the image and used to SUBTRACT rows from the frame's progress word -- the per-frame
    kernel then waited for a count that never came (RMCV_FRAME_TIMEOUT after 2 s).  Forked (one stream) and explicit form."""
    import torch
    fr = synth.batch(52000 + h, n, w, h, CAMP_BLUE, 1, threads=16)
    refs = [oracle.detect_frame(f) for f in fr]
    c = Context(device=0, max_frames=8, max_width=w, max_height=h)
    c.set_option(OPT_HANDOVER, 1)
    c.upload(fr)
    for explicit in (False, True, False):
        if explicit:
            sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
            c.run(default_params(), STAGE_BINARY, sa.cuda_stream)
            c.run(default_params(), (STAGE_ALL & ~STAGE_BINARY) | STAGE_HANDOVER, sb.cuda_stream)
        else:
            c.run(default_params(), STAGE_ALL)
        c.sync()
        torch.cuda.synchronize()
        assert not c.counts()["status"].any()
        arm, offs = c.armours()
        for f in range(n):
            assert np.array_equal(c.binary(f), refs[f]["binary"]), (explicit, f)
            assert arm[offs[f]:offs[f + 1]].tobytes() == refs[f]["armours"].tobytes(), (explicit, f)
    c.close()


def test_handover_after_a_pixel_kernel_that_did_not_publish(oracle):
    """rmcv_extract_color launches the pixel kernel WITHOUT progress words; a RMCV_STAGE_HANDOVER run behind it must
    not wait for a label nothing writes (it used to block its stream for good) -- it is refused: there is no batch pixel kernel to follow"""
    w, h = 1280, 1024
    fr = synth.batch(63000, 1, w, h, CAMP_BLUE, 0, threads=4)
    c = Context(device=0, max_frames=1, max_width=w, max_height=h)
    c.set_option(OPT_HANDOVER, 1)
    c.upload(fr)
    c.run(default_params(), STAGE_BINARY)
    c.sync()
    c.extract_color_csr(fr[0])
    with pytest.raises(RmcvError):
        c.run(default_params(), (STAGE_ALL & ~STAGE_BINARY) | STAGE_HANDOVER)
    c.upload(fr)
    c.run(default_params(), STAGE_ALL)                     # and the context goes on working
    c.sync()
    arm, offs = c.armours()
    assert arm.tobytes() == oracle.detect_frame(fr[0])["armours"].tobytes()
    c.close()
