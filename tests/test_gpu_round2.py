"""GPU parity, round 2: the BASELINE configurations at their FULL sizes that round 1 only covered at test size (C2 red/dilate,
C5 identities), the per-frame drop-in chain with its device-resident hand-over, and the robustness contracts of the context
(stream changes, stale HIP errors on the thread, status bits across split runs).  Everything goes through the C-ABI; the oracle
is the checker."""
import ctypes
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from rmcv_amd import (CAMP_BLUE, CAMP_RED, MORPH_CLOSE, MORPH_DILATE, OPT_FRAME_UPLOAD, OPT_IMAGE_EXPORT, OPT_PIXEL_GROUPS, STAGE_ALL, STAGE_ARMOURS,
                      STAGE_BINARY, STAGE_BLOBS, STAGE_CONTOURS, STAGE_IDENTITY, Context, RmcvError, default_params, synth)

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- BASELINE config 2 at full size
@pytest.mark.parametrize("morph,groups", [(MORPH_DILATE, 2), (MORPH_CLOSE, 4), (MORPH_DILATE, 4)])
def test_c2_full_size_red_binary_every_frame(oracle, morph, groups):
    """configs[1]: batch 256 x 1280x1024, red team (mirrored stream), channel subtract + threshold + dilate (and close), pixel
    kernel only -- k_binary<2,0,...> with the tapered strip queue, which bench.py's c2_binary_only times.  Every frame's
    `binary` against the oracle's restatement of src/imgproc.cpp:52-69."""
    n = 256
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    c.set_option(OPT_PIXEL_GROUPS, groups)
    frames = synth.batch(90000, n, 1280, 1024, CAMP_RED, 1, threads=16)
    p = default_params(camp=CAMP_RED, morph=morph)
    c.upload(frames)
    for _ in range(3):                       # consecutive launches of one context: the strip queue must restart cleanly each time
        c.run(p, STAGE_BINARY)
    c.sync()
    with ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(lambda f: oracle.extract_binary(frames[f], CAMP_RED, 80, morph), range(n)))
    lit = 0
    for f in range(n):
        assert np.array_equal(c.binary(f), refs[f]), (f, morph, groups)
        lit += int(np.count_nonzero(refs[f]))
    assert lit > 1000 * n                    # the red stream really has foreground
    c.close()


# ---------------------------------------------------------------- BASELINE config 5 at full size
def test_c5_full_size_identities_every_frame(oracle):
    """configs[4]: batch 256 x 1920x1200, full path + icon rectification + SVM (executable/main.cpp:178-181): armours (with
    the clamped icon vertices), identities and the 20x20 icons of every frame"""
    svm = synth.svm_weights()
    n = 256
    c = Context(device=0, max_frames=n, max_width=1920, max_height=1200)
    c.svm_load(*svm)
    frames = synth.batch(120000, n, 1920, 1200, CAMP_BLUE, 0, threads=16)
    c.upload(frames)
    c.run(default_params(), STAGE_ALL | STAGE_IDENTITY)
    c.sync()
    arm, offs = c.armours()
    ident = c.identities()
    assert len(ident) == len(arm) and len(arm) > n

    def ref(f):
        r = oracle.detect_frame(frames[f])
        return oracle.classify_armours(frames[f], r["armours"], svm)
    with ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(ref, range(n)))
    seen = set()
    for f in range(n):
        ri, ra, ricons = refs[f]
        assert arm[offs[f]:offs[f + 1]].tobytes() == ra.tobytes(), f
        assert np.array_equal(ident[offs[f]:offs[f + 1]], ri), f
        assert np.array_equal(c.icons(f), ricons), f
        seen |= set(ri.tolist())
    assert len(seen) >= 3                    # the stand-in weights do separate the icons into several classes
    c.close()


# ---------------------------------------------------------------- the per-frame drop-in chain
def chain(ctx, img, camp=CAMP_BLUE, lb=80, morph=MORPH_CLOSE):
    pts, offs, binary = ctx.extract_color_csr(img, camp, lb, morph)
    blobs, src, neg = ctx.filter_lightblobs(pts, offs, enemy=camp)
    arm = ctx.filter_armours(blobs, enemy=camp)
    return pts, offs, binary, blobs, src, neg, arm


def check_chain(oracle, got, img, camp=CAMP_BLUE):
    pts, offs, binary, blobs, src, neg, arm = got
    p = oracle.default_params(camp=camp)
    ref = oracle.detect_frame(img, p)
    assert np.array_equal(binary, ref["binary"])
    assert np.array_equal(offs, ref["offs"]) and np.array_equal(pts, ref["pts"])
    rb, rs, rn = oracle.filter_lightblobs(ref["pts"], ref["offs"], p)
    assert blobs.tobytes() == rb.tobytes() and np.array_equal(src, rs) and np.array_equal(neg, rn)
    assert arm.tobytes() == ref["armours"].tobytes()
    return ref


@pytest.mark.parametrize("upload,image_export", [(1, 0), (0, 0), (2, 0), (0, 1), (2, 1)])
def test_per_frame_chain_equals_oracle(oracle, upload, image_export):
    """rm::extract_color -> rm::filter_lightblobs -> rm::filter_armours as executable/main.cpp:172-176 calls them, one host frame
    at a time: results stay on the device between the calls (resident hand-over); all three upload modes"""
    c = Context(device=0, max_frames=1, max_width=1920, max_height=1200)
    c.set_option(OPT_FRAME_UPLOAD, upload)
    c.set_option(OPT_IMAGE_EXPORT, image_export)                # the byte image through the runtime's pageable copy / the library's export kernel
    total = 0
    buf = np.empty((1024, 1280, 3), np.uint8)                   # one reused host buffer, as a camera ring would be
    for idx in (0, 1, 2, 3, 1004, 1005):
        buf[:] = synth.frame(idx, 1280, 1024, CAMP_BLUE, 1 if idx > 1000 else 0)
        total += len(check_chain(oracle, chain(c, buf), buf)["armours"])
    assert total > 6
    keep = []                                                   # mode 2 pins the caller's buffers in place: they must outlive the context
    for idx, (w, h) in [(5, (1920, 1200)), (6, (640, 480)), (7, (333, 200))]:   # geometry changes, unaligned widths
        img = synth.frame(idx, w, h)
        keep.append(img)
        check_chain(oracle, chain(c, img), img)
    red = synth.frame(11, 1280, 1024, CAMP_RED, 0)
    keep.append(red)
    check_chain(oracle, chain(c, red, CAMP_RED), red, CAMP_RED)
    c.close()
    del keep


def test_per_frame_handover_only_when_bytes_match(ctx, oracle):
    """the resident buffers are used only for exactly the bytes returned; anything else is uploaded and gives ITS result"""
    img = synth.frame(21, 1280, 1024, CAMP_BLUE, 1)
    pts, offs, binary = ctx.extract_color_csr(img)
    p = oracle.default_params()
    # (a) a caller that drops the first contour before filtering
    pts2, offs2 = pts[offs[1]:].copy(), (offs[1:] - offs[1]).astype(np.int32)
    b2, s2, n2 = ctx.filter_lightblobs(pts2, offs2)
    rb, rs, rn = oracle.filter_lightblobs(pts2, offs2, p)
    assert b2.tobytes() == rb.tobytes() and np.array_equal(s2, rs) and np.array_equal(n2, rn)
    # (b) same sizes, one coordinate changed
    pts, offs, binary = ctx.extract_color_csr(img)
    pts3 = pts.copy()
    k = int(np.argmax(np.diff(offs)))                            # the longest contour
    pts3["x"][offs[k]] += 1
    b3, s3, n3 = ctx.filter_lightblobs(pts3, offs)
    rb, rs, rn = oracle.filter_lightblobs(pts3, offs, p)
    assert b3.tobytes() == rb.tobytes() and np.array_equal(n3, rn)
    # (c) the blob list: reordered by the caller -> uploaded; then the unchanged list twice (labeler.cpp calls it with two parameter sets)
    pts, offs, _ = ctx.extract_color_csr(img)
    blobs, _, _ = ctx.filter_lightblobs(pts, offs)
    assert len(blobs) >= 2
    rev = blobs[::-1].copy()
    assert ctx.filter_armours(rev).tobytes() == oracle.filter_armours(rev, p).tobytes()
    blobs, _, _ = ctx.filter_lightblobs(pts, offs)
    loose = oracle.default_params(angle_diff_max=999.0, shear_max=999.0, length_ratio_max=0.01)   # labeler.cpp:75-82
    assert ctx.filter_armours(blobs).tobytes() == oracle.filter_armours(blobs, p).tobytes()
    assert ctx.filter_armours(blobs, 999.0, 999.0, 0.01).tobytes() == oracle.filter_armours(blobs, loose).tobytes()
    # (d) other parameters on resident contours (second call of the chain with a different gate)
    b4, _, n4 = ctx.filter_lightblobs(pts, offs, tilt_max=20.0, ratio_range=(3.0, 10.0))
    p4 = oracle.default_params(tilt_max=20.0, ratio_lo=3.0, ratio_hi=10.0)
    rb, _, rn = oracle.filter_lightblobs(pts, offs, p4)
    assert b4.tobytes() == rb.tobytes() and np.array_equal(n4, rn)


def test_per_frame_results_beyond_the_copy_windows(oracle):
    """more contours / points / blobs / armours than the speculative download windows hold (1024 / 8192 / 64 / 32)"""
    c = Context(device=0, max_frames=1, max_width=1280, max_height=1024, max_contours=8192, max_points=1 << 18,
                max_blobs=1024, max_armours=4096)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 40, (1024, 1280, 3), dtype=np.uint8)
    img[rng.random((1024, 1280)) < 0.004] = (255, 200, 0)        # ~5000 specks: contours beyond the offs window
    k = 0
    for y in range(40, 1000, 60):                                 # a grid of upright bars: > 64 blobs, > 32 armours
        for x in range(30, 1250, 28):
            img[y:y + 36, x:x + 5] = (255, 190, 10)
            k += 1
    got = chain(c, img)
    ref = check_chain(oracle, got, img)
    assert len(ref["offs"]) - 1 > 1024 and len(ref["pts"]) > 8192 and len(ref["blobs"]) > 64 and len(ref["armours"]) > 32
    # output capacity smaller than the result: reported, with the needed counts
    import ctypes as C
    from rmcv_amd.abi import POINT, lib, ptr
    pts = np.empty(16, POINT)
    offs = np.empty(9, np.int32)
    nc, npnt = C.c_int32(0), C.c_int32(0)
    rc = lib().rmcv_extract_color(c._h, ptr(img), 1280, 1024, 3 * 1280, CAMP_BLUE, 80, MORPH_CLOSE, None, ptr(pts), 16, ptr(offs), 8,
                                  C.byref(nc), C.byref(npnt))
    assert rc == -2 and nc.value == len(ref["offs"]) - 1 and npnt.value == len(ref["pts"])
    c.close()


# ---------------------------------------------------------------- robustness of a context
def test_two_streams_on_one_context_are_ordered(oracle):
    """launches of one context share its buffers and k_binary's strip queue: handing in a different stream for every call must
    order them (an event chain inside the library), not race"""
    import torch
    n = 32
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    fa = synth.batch(300, n, 1280, 1024, CAMP_BLUE, 0, threads=16)
    fb = synth.batch(400, n, 1280, 1024, CAMP_BLUE, 1, threads=16)
    ta, tb = torch.from_numpy(fa).cuda(), torch.from_numpy(fb).cuda()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    p = default_params()
    for rep in range(6):                                          # alternate streams AND inputs without any host sync in between
        t, s = (ta, s1) if rep % 2 == 0 else (tb, s2)
        c.bind_device_frames(t.data_ptr(), n, 1024, 1280, keepalive=t)
        c.run(p, STAGE_BINARY | STAGE_CONTOURS, s.cuda_stream)
        c.run(p, STAGE_BLOBS | STAGE_ARMOURS, (s2 if s is s1 else s1).cuda_stream)   # and split one step over both
    c.sync()
    arm, offs = c.armours()                                       # the last pass ran on fb
    for f in range(n):
        ref = oracle.detect_frame(fb[f])
        assert np.array_equal(c.binary(f), ref["binary"]), f
        assert arm[offs[f]:offs[f + 1]].tobytes() == ref["armours"].tobytes(), f
    c.close()


def test_stale_hip_error_on_the_thread_does_not_fail_or_derail_a_run(ctx, oracle):
    """an application may have handled a failed HIP call by return code (here: an absurd hipMalloc); the thread's sticky
    last-error slot must neither make the next launch look failed nor put k_binary's strip queue out of step (round 1 read
    hipGetLastError after the launch and advanced a host-side mirror of the queue only on success)"""
    hip = ctypes.CDLL("libamdhip64.so")
    frames = synth.batch(800, 4, 1280, 1024, CAMP_BLUE, 0)
    ctx.upload(frames)
    p = default_params()
    for rep in range(3):
        q = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t(1 << 60)) != 0      # fails, leaves the sticky error behind
        ctx.run(p, STAGE_ALL)                                                      # raises if the run reports a failure
        ctx.sync()
        arm, offs = ctx.armours()
        for f in range(4):
            ref = oracle.detect_frame(frames[f])
            assert np.array_equal(ctx.binary(f), ref["binary"]), (rep, f)
            assert arm[offs[f]:offs[f + 1]].tobytes() == ref["armours"].tobytes(), (rep, f)
    hip.hipGetLastError()   # reading the slot resets it: torch (same runtime, same thread) checks it after its own calls


def test_bad_call_leaves_the_context_usable(ctx, oracle):
    frames = synth.batch(810, 2, 1280, 1024)
    ctx.upload(frames)
    with pytest.raises(RmcvError):
        ctx.run(default_params(), 0)                 # bad stage mask
    with pytest.raises(RmcvError):
        ctx.run(default_params(morph=7), STAGE_ALL)  # bad morph
    arm, offs = ctx.detect_batch(frames)
    for f in range(2):
        assert arm[offs[f]:offs[f + 1]].tobytes() == oracle.detect_frame(frames[f])["armours"].tobytes()


def test_status_bits_survive_a_split_run(oracle):
    """run(BINARY|CONTOURS) then run(BLOBS|ARMOURS): the contour stage's overflow bits must still be reported afterwards"""
    c = Context(device=0, max_frames=2, max_width=640, max_height=480, max_contours=4)
    frames = synth.batch(77, 2, 640, 480, CAMP_BLUE, 1)
    frames[1][::7, ::9] = (255, 200, 0)              # hundreds of specks: more than 4 contours
    c.upload(frames)
    c.run(default_params(), STAGE_BINARY | STAGE_CONTOURS)
    c.run(default_params(), STAGE_BLOBS | STAGE_ARMOURS)
    c.sync()
    st = c.counts()["status"]
    assert st[1] & 1, st                             # RMCV_FRAME_OVF_CONTOURS
    with pytest.raises(RmcvError) as e:
        c.armours()
    assert e.value.code == -2
    c.close()


# ---------------------------------------------------------------- optional: real OpenCV, if the box happens to have it
def test_oracle_against_real_opencv_if_present(oracle):
    """the oracle is pinned by the build's own KATs only, because OpenCV exists neither in the reference tree nor in
    this image.  If a box does have cv2, cross-check the integer stages exactly and the ellipse fit to float rounding.
    Skips cleanly otherwise; never required."""
    cv2 = pytest.importorskip("cv2")
    for idx in (0, 1, 1002):
        img = synth.frame(idx, 1280, 1024, CAMP_BLUE, 1 if idx > 1000 else 0)
        ref = oracle.detect_frame(img)
        b, g, r = cv2.split(img)
        binary = cv2.inRange(cv2.subtract(b, r), 80, 255)
        binary = cv2.morphologyEx(binary, cv2.MORPH_CLOSE, cv2.getStructuringElement(cv2.MORPH_RECT, (3, 3)))
        assert np.array_equal(binary, ref["binary"])
        cs, _ = cv2.findContours(binary, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_NONE)
        assert len(cs) == len(ref["offs"]) - 1
        for i, cnt in enumerate(cs):
            mine = ref["pts"][ref["offs"][i]:ref["offs"][i + 1]]
            assert np.array_equal(cnt.reshape(-1, 2), np.stack([mine["x"], mine["y"]], 1)), (idx, i)
            if len(cnt) >= 6:
                (cx, cy), (w, h), ang = cv2.fitEllipseDirect(cnt)
                e, _ = oracle.fit_ellipse_direct(mine)
                assert np.allclose([cx, cy, w, h, ang], [e["cx"], e["cy"], e["w"], e["h"], e["angle"]], rtol=1e-5, atol=1e-4), (idx, i)
                (mx, my), (mw, mh), ma = cv2.minAreaRect(cnt)
                m = oracle.min_area_rect(mine)
                assert np.allclose(sorted([mw, mh]), sorted([m["w"], m["h"]]), rtol=1e-5, atol=1e-4), (idx, i)


# ---------------------------------------------------------------- the C-ABI gather (RCCL called by the library), one rank
def test_abi_gather_single_rank_moves_the_record(oracle):
    """rmcv_comm_* + rmcv_gather with a group of one (a one-GPU box): the root's record must arrive unchanged in slot 0 of the
    receive buffer, asynchronously on the caller's stream -- and decode to the batch's armour list"""
    import ctypes as C

    import torch

    from rmcv_amd import dist as rdist
    from rmcv_amd.abi import COMM_ID_BYTES, lib
    L = lib()
    idb = (C.c_uint8 * COMM_ID_BYTES)()
    assert L.rmcv_comm_unique_id(idb) == 0
    h = C.c_void_p()
    assert L.rmcv_comm_create(idb, 1, 0, 0, C.byref(h)) == 0
    n, cap = 8, 64
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    frames = synth.batch(4000, n, 1280, 1024, CAMP_BLUE, 1)
    c.upload(frames)
    s = torch.cuda.Stream()
    head, total = rdist.record_layout(n, cap)
    rec = rdist.new_record(n, cap, torch.device("cuda", 0))
    recv = torch.full((total,), 0xEE, dtype=torch.uint8, device="cuda")
    c.run(default_params(), STAGE_ALL, s.cuda_stream)
    c.compact_armours_into(rec.data_ptr() + head, cap, rec.data_ptr(), s.cuda_stream)
    assert L.rmcv_gather(h, C.c_void_p(rec.data_ptr()), C.c_int64(total), C.c_void_p(recv.data_ptr()), 0, C.c_void_p(s.cuda_stream)) == 0
    s.synchronize()
    assert torch.equal(recv, rec)
    arm, offs = rdist.unpack_records([recv], n, cap)
    ref = [oracle.detect_frame(frames[f])["armours"] for f in range(n)]
    assert offs.tolist() == np.cumsum([0] + [len(r) for r in ref]).tolist()
    assert arm.tobytes() == np.concatenate(ref).tobytes() and len(arm) > 0
    # bad arguments on a live communicator
    assert L.rmcv_gather(h, C.c_void_p(rec.data_ptr()), C.c_int64(total), None, 0, None) == -1      # root without a receive buffer
    assert L.rmcv_gather(h, C.c_void_p(rec.data_ptr()), C.c_int64(total), C.c_void_p(recv.data_ptr()), 1, None) == -1   # no such root
    L.rmcv_comm_destroy.restype = None
    L.rmcv_comm_destroy.argtypes = [C.c_void_p]
    L.rmcv_comm_destroy(h)
    c.close()


def test_run_ahead_changes_nothing_but_the_timing(oracle):
    """RMCV_OPT_RUN_AHEAD: rmcv_extract_color enqueues the filters with the previous frame's parameters; the filter calls hand
    the results over when this frame asks for the same.  Same results with it on and off, with parameters that stay, change and
    change back, and when a caller edits the lists in between"""
    from rmcv_amd import OPT_RUN_AHEAD
    frames = [synth.frame(i, 1280, 1024, CAMP_BLUE, i % 2) for i in range(30, 38)]
    gates = [dict(), dict(), dict(tilt_max=20.0, ratio_range=(3.0, 10.0)), dict(), dict(), dict(area_range=(50.0, 500.0)), dict(), dict()]
    pairs = [dict(), dict(), dict(), dict(shear_max=40.0), dict(), dict(), dict(), dict()]
    out = {}
    for mode in (1, 0):
        c = Context(device=0, max_frames=1, max_width=1280, max_height=1024)
        c.set_option(OPT_RUN_AHEAD, mode)
        res = []
        for k, f in enumerate(frames):
            pts, offs, binary = c.extract_color_csr(f)
            if k == 6:                                         # a caller that drops the last contour: nothing run ahead may be used
                pts, offs = pts[:offs[-2]].copy(), offs[:-1].copy()
            blobs, src, neg = c.filter_lightblobs(pts, offs, **gates[k])
            arm = c.filter_armours(blobs, **pairs[k])
            res.append((pts.tobytes(), offs.tobytes(), binary.tobytes(), blobs.tobytes(), src.tobytes(), neg.tobytes(), arm.tobytes()))
            # against the oracle with the same parameters
            g = gates[k]
            p = oracle.default_params(tilt_max=g.get("tilt_max", 70.0), ratio_lo=g.get("ratio_range", (1.5, 80.0))[0],
                                      ratio_hi=g.get("ratio_range", (1.5, 80.0))[1], area_lo=g.get("area_range", (10.0, 99999.0))[0],
                                      area_hi=g.get("area_range", (10.0, 99999.0))[1], shear_max=pairs[k].get("shear_max", 22.0))
            rb, rs, rn = oracle.filter_lightblobs(pts, offs, p)
            assert blobs.tobytes() == rb.tobytes() and np.array_equal(neg, rn), (mode, k)
            assert arm.tobytes() == oracle.filter_armours(rb, p).tobytes(), (mode, k)
        out[mode] = res
        c.close()
    assert out[0] == out[1]


def test_batch_beyond_the_32_bit_extent_keeps_the_fast_path(oracle):
    """1100 frames of 1280x1024 are 4.3 GB of input: more than one launch of the pixel kernel can address with its 32-bit buffer
    offsets.  The library splits such a batch into launches over frame ranges (same stream, pointers advanced); every stage of a
    sample of frames -- the first, the ones around the split, the last -- against the oracle"""
    n = 1100
    lim = 0xFFFFFF00
    split = (lim - 1) // (3 * 1280 * 1024)                       # frames per launch (the input is the largest extent)
    assert 1 < split < n
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024, max_points=16384, max_contours=512)
    frames = synth.batch(300000, n, 1280, 1024, CAMP_BLUE, 1, threads=16)
    arm, offs = c.detect_batch(frames)
    assert not (c.counts()["status"] & 15).any()
    for f in (0, 1, split - 1, split, split + 1, 2 * split - 1 if 2 * split - 1 < n else n - 2, n - 1):
        ref = oracle.detect_frame(frames[f])
        assert np.array_equal(c.binary(f), ref["binary"]), f
        pts, co = c.contours(f)
        assert np.array_equal(co, ref["offs"]) and np.array_equal(pts, ref["pts"]), f
        assert arm[offs[f]:offs[f + 1]].tobytes() == ref["armours"].tobytes(), f
    c.close()


@pytest.mark.parametrize("morph", [0, 1, 2])
def test_pixel_kernel_coalesced_loader_geometries(oracle, morph):
    """The wave-coalesced loader of k_binary (256-pixel blocks of four rows per wavefront, 12 bytes per lane): widths that are a
    multiple of 64 but not of 256 (ragged last block), one block, many blocks; heights below, at and around the 32-row strip and
    its row quads (1, 2, 3, 5, 31..34, 63..65); packed rows and rows / frames with padding; every camp, lb at the borders
    (lb <= 0 passes everything, 256 nothing).  Byte image of every frame against the oracle."""
    import torch
    from rmcv_amd import (CAMP_BLUE, CAMP_GUIDELIGHT, CAMP_NEUTRAL, CAMP_RED, STAGE_BINARY, Context, default_params)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4242 + morph)
    # (a last block of at most 128 pixels -- 64, 128, 320, 384, 576, 640, 832, 1344, 1920 -- shares its wavefront between TWO row quads)
    widths = [64, 128, 192, 256, 320, 448, 512, 576, 768, 832, 1280, 1344, 384, 640, 1920]
    heights = [1, 2, 3, 5, 31, 32, 33, 34, 63, 64, 65, 100, 7, 36, 41]
    cases = [(widths[i % len(widths)], heights[(i * 5 + morph) % len(heights)]) for i in range(30)]
    for ci, (w, h) in enumerate(cases):
        n = 1 + ci % 3
        pad_row, pad_frame = [(0, 0), (16, 0), (48, 64), (0, 4096)][ci % 4]
        stride = 3 * w + pad_row
        pitch = stride * h + pad_frame
        host = rng.integers(0, 256, n * pitch, dtype=np.uint8)            # padding bytes are random too: they must not matter
        frames = np.empty((n, h, w, 3), np.uint8)
        for f in range(n):
            img = rng.integers(0, 48, (h, w, 3), dtype=np.uint8)
            m = rng.random((h, w)) < 0.35                                  # dense foreground: exercises every morphology border
            img[m] = (255, 120, 10) if ci % 2 == 0 else (10, 200, 255)
            img[rng.random((h, w)) < 0.05] = (200, 255, 40)
            frames[f] = img
            for y in range(h):
                host[f * pitch + y * stride:f * pitch + y * stride + 3 * w] = img[y].reshape(-1)
        buf = torch.from_numpy(host).to(dev)
        c = Context(device=0, max_frames=n, max_width=w, max_height=h)
        c.bind_device_frames(buf.data_ptr(), n, h, w, stride=stride, frame_pitch=pitch, keepalive=buf)
        for camp, lb in [(CAMP_BLUE, 80), (CAMP_RED, 80), (CAMP_GUIDELIGHT, 60), (CAMP_NEUTRAL, 1), (CAMP_BLUE, 0), (CAMP_RED, 256)]:
            c.run(default_params(camp=camp, lower_bound=lb, morph=morph), STAGE_BINARY)
            c.sync()
            for f in range(n):
                ref = oracle.extract_binary(frames[f], camp, lb, morph)
                assert np.array_equal(c.binary(f), ref), (w, h, n, stride, pitch, camp, lb, f)
        c.close()


def test_pixel_kernel_very_wide_frames(oracle):
    """frames wider than ~6700 pixels need more than 64 KiB of dynamic LDS for the strip's two bit planes: the launch raises the
    kernel's limit instead of failing (7168 and 8192 wide, both loaders)"""
    for (w, h, stride) in [(7168, 40, 3 * 7168), (8192, 37, 3 * 8192), (7200, 35, 3 * 7200 + 5)]:
        rng = np.random.default_rng(w + h)
        img = rng.integers(0, 48, (h, w, 3), dtype=np.uint8)
        img[rng.random((h, w)) < 0.3] = (255, 100, 10)
        c = Context(device=0, max_frames=1, max_width=w, max_height=h, max_contours=1 << 16, max_points=1 << 20)
        if stride == 3 * w:
            c.upload(img[None])
        else:
            import torch
            host = np.zeros(stride * h, np.uint8)
            for y in range(h):
                host[y * stride:y * stride + 3 * w] = img[y].reshape(-1)
            buf = torch.from_numpy(host).to(torch.device("cuda", 0))
            c.bind_device_frames(buf.data_ptr(), 1, h, w, stride=stride, keepalive=buf)
        c.run(default_params(), STAGE_BINARY)
        c.sync()
        assert np.array_equal(c.binary(0), oracle.extract_binary(img, CAMP_BLUE, 80, MORPH_CLOSE)), (w, h)
        c.close()


def test_armour_list_compaction_many_frames_and_overflow(oracle):
    """rmcv_batch_compact_armours (the record of the multi-GPU gather): 300 frames -- more than one chunk of the count scan, 19
    workgroups -- with wide-open gates (0 to dozens of armours per frame); the frame-major list and its offsets equal the per-frame
    lists laid end to end, and a list capacity below the total drops exactly the armours beyond it"""
    import torch
    n, w, h = 300, 320, 192
    frames = np.stack([synth.frame(7000 + i, w, h, CAMP_BLUE, i % 2) for i in range(n)])
    frames[5] = 0                                                   # a frame with nothing in it
    p = default_params()
    p.tilt_max, p.ratio_lo, p.ratio_hi, p.area_lo, p.area_hi = 1e9, 0.0, 1e30, 0.0, 1e30
    p.angle_diff_max, p.shear_max, p.length_ratio_max = 1e9, 1e9, 0.0
    c = Context(device=0, max_frames=n, max_width=w, max_height=h, max_blobs=64, max_armours=512)
    c.upload(frames)
    c.run(p, STAGE_ALL)
    c.sync()
    arm, offs = c.armours()                                         # host-side reference: the per-frame slots, read back one by one
    total = int(offs[-1])
    assert total > 600 and offs[6] == offs[5]
    dt = arm.dtype
    for cap in (total + 10, total, total // 2 + 3, 1):
        out = torch.full((cap * dt.itemsize,), 0xEE, dtype=torch.uint8, device="cuda")
        fo = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        c.compact_armours_into(out.data_ptr(), cap, fo.data_ptr())
        torch.cuda.synchronize()
        assert fo.cpu().numpy().tolist() == offs.tolist(), cap      # the offsets always describe the complete list
        got = np.frombuffer(out.cpu().numpy().tobytes(), dtype=dt)
        keep = min(cap, total)
        assert got[:keep].tobytes() == arm[:keep].tobytes(), cap
    c.close()


@pytest.mark.parametrize("camp_name,lb,morph", [("GUIDELIGHT", 60, MORPH_CLOSE), ("NEUTRAL", 80, MORPH_DILATE), ("BLUE", 1, MORPH_CLOSE),
                                                ("BLUE", 255, 0)])
def test_full_size_binary_other_camps_and_bounds(oracle, camp_name, lb, morph):
    """the other two instantiations of the pixel kernel (G - R for the guide light; NEUTRAL = R - B, imgproc.cpp:56-65) and the
    ends of the lower bound's range (1: nearly everything above the noise floor passes -- dense planes; 255: only saturated
    differences) on 64 full-size frames each"""
    import rmcv_amd
    camp = getattr(rmcv_amd, "CAMP_" + camp_name)
    n = 64
    frames = synth.batch(130000 + lb, n, 1280, 1024, CAMP_BLUE, 1, threads=16)
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    c.upload(frames)
    c.run(default_params(camp=camp, lower_bound=lb, morph=morph), STAGE_BINARY)
    c.sync()
    with ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(lambda f: oracle.extract_binary(frames[f], camp, lb, morph), range(n)))
    for f in range(n):
        assert np.array_equal(c.binary(f), refs[f]), (f, camp_name, lb, morph)
    c.close()
