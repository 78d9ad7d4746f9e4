"""Canaries around every device buffer of a context (rmcv_ctx_check_guards): no kernel of the path stores outside its buffers.

Round 2 lost one GPU run (gpurun_out/r2d: rc 134, a silent abort inside the first 1920x1200 launch while k_binary was being
rewritten to unconditional raw-buffer loads/stores) without a recorded cause.  This geometry is the first in the suite whose strips
do not divide the frame (37.5 strips of 32 rows: the last strip's rows 16..31 lie below the image, halo rows included), whose rows end
in a ragged 256-pixel block (7.5 blocks), and -- with the 6 frames of the test that aborted -- whose strip count (228) is not a
multiple of the 8 XCD queues.  Every store such a strip could misplace (bit-plane words and row masks of rows >= h, bytes of the
image beyond the last row, LDS spill-over into the neighbouring frame's rows) lands either in a neighbouring frame -- the parity
assertions below -- or, for the last frame of a context sized exactly for the batch, in a guard zone.  DESIGN.md section 9 has
the audit."""
import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,n", [(1920, 1200, 6), (1280, 1024, 3), (1984, 1210, 2), (1000, 700, 2), (64, 64, 5)])
def test_no_store_outside_the_buffers(oracle, w, h, n):
    from rmcv_amd import CAMP_RED, MORPH_DILATE, STAGE_ALL, STAGE_BINARY, STAGE_IDENTITY, Context, default_params, synth
    ctx = Context(device=0, max_frames=n, max_width=w, max_height=h)     # sized EXACTLY: the last frame ends at the rear guards
    assert ctx.check_guards()[0] == 0
    frames = synth.batch(90, n, w, h, variant=1)
    if w == 1920:
        ctx.svm_load(*synth.svm_weights())
    ctx.upload(frames)
    ctx.run(default_params(), STAGE_ALL | (STAGE_IDENTITY if w == 1920 else 0))
    ctx.sync()
    bad, what = ctx.check_guards()
    assert bad == 0, what
    for f in (0, n - 1):                                                  # the frames whose neighbours are a guard zone
        ref = oracle.detect_frame(frames[f])
        assert np.array_equal(ctx.binary(f), ref["binary"]), f
        pts, co = ctx.contours(f)
        assert np.array_equal(co, ref["offs"]) and np.array_equal(pts, ref["pts"]), f
    arm, offs = ctx.armours()
    ref_last = oracle.detect_frame(frames[n - 1])
    assert arm[offs[n - 1]:offs[n]].tobytes() == ref_last["armours"].tobytes()
    # the other instantiations of the pixel kernel (red: channels swapped; dilate: one halo row) and a lower bound that passes everything
    for p in (default_params(camp=CAMP_RED, morph=MORPH_DILATE), default_params(lower_bound=0)):
        ctx.run(p, STAGE_BINARY)
        ctx.sync()
        bad, what = ctx.check_guards()
        assert bad == 0, what
    ctx.close()


@pytest.mark.gpu
def test_guards_notice_a_stray_store():
    """the canary itself: four bytes written behind a buffer's end are reported with the buffer's name.  The stray store is made
    with the ABI's own compaction call, told (wrongly, on purpose) that the context's one-entry `n_armours` array has room for
    the two offsets of a one-frame batch."""
    import torch
    from rmcv_amd import STAGE_ALL, Context, default_params, synth
    ctx = Context(device=0, max_frames=1, max_width=128, max_height=64)
    ctx.upload(synth.batch(3, 1, 128, 64))
    ctx.run(default_params(), STAGE_ALL)
    ctx.sync()
    assert ctx.check_guards()[0] == 0
    _, d_counts, _, n = ctx.device_views()
    assert n == 1
    sink = torch.zeros(88 * 16, dtype=torch.uint8, device="cuda:0")
    ctx.compact_armours_into(sink.data_ptr(), 16, d_counts)             # frame_offs[1] lands 4 bytes past the end of n_armours
    ctx.sync()
    torch.cuda.synchronize()
    bad, what = ctx.check_guards()
    assert bad == 1 and "n_armours" in what and "behind" in what, what
    ctx.close()
