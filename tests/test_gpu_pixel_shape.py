"""RMCV_OPT_PIXEL_SHAPE 1: k_binary_ws (rmcv_amd/csrc/k_binary_ws.inc), the wave-specialised pixel kernel for whole batches -- loader
wavefronts a strip ahead of storer wavefronts, dilate and erode fused per output word.  Same stage of the reference
(src/imgproc.cpp:52-69: split, channel subtract, inRange, 3x3 close), so the same bar: byte image, contours, blobs and armours of
every frame equal the oracle's, for every camp, morph and geometry the kernel takes; whatever it does not take falls back to
k_binary (and the diagnostic counter says which of the two ran)."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from rmcv_amd import (CAMP_BLUE, CAMP_GUIDELIGHT, CAMP_RED, MORPH_CLOSE, MORPH_DILATE, MORPH_NONE, OPT_PIXEL_SHAPE, STAGE_ALL, Context,
                      default_params, synth)
from rmcv_amd import abi

pytestmark = pytest.mark.gpu


def ws_launches():
    return abi.lib().rmcv_pixel_ws_launches()


def make_frames(seed, n, w, h):
    """the synthetic scenes where the generator takes the geometry; bars, specks and noise drawn here where it does not (tiny heights)"""
    try:
        frames = synth.batch(seed, n, w, h, CAMP_BLUE, 0, threads=16)
        frames[1::3] = synth.batch(seed + 2000, len(frames[1::3]), w, h, CAMP_BLUE, 1, threads=16)     # (the noisy variant)
        return frames
    except ValueError:
        rng = np.random.default_rng(seed)
        frames = rng.integers(0, 60, (n, h, w, 3), dtype=np.uint8)
        for f in range(n):
            for _ in range(6):
                x, y = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 3))
                bw, bh = int(rng.integers(2, 24)), int(rng.integers(2, h))
                frames[f, y:y + bh, x:x + bw, 0] = 255
                frames[f, y:y + bh, x:x + bw, 2] = 10
            speck = rng.random((h, w)) < 0.01
            frames[f, speck, 0] = 250
            frames[f, speck, 2] = 0
        frames[:, 0, :, 0], frames[:, 0, :, 2] = 255, 0          # the image's first and last row and column lit: the border rules
        frames[:, -1, :, 0], frames[:, -1, :, 2] = 255, 0
        frames[:, :, 0, 0], frames[:, :, 0, 2] = 255, 0
        frames[:, :, -1, 0], frames[:, :, -1, 2] = 255, 0
        return frames


def run_and_check(oracle, frames, camp, morph, lb=80, expect_ws=True, stride_pad=0):
    n, h, w, _ = frames.shape
    c = Context(device=0, max_frames=n, max_width=w, max_height=h)
    c.set_option(OPT_PIXEL_SHAPE, 1)
    p = default_params()
    p.camp, p.morph, p.lower_bound = camp, morph, lb
    if stride_pad:
        import torch
        padded = np.zeros((n, h, 3 * w + stride_pad), np.uint8)
        padded[:, :, :3 * w] = frames.reshape(n, h, 3 * w)
        d = torch.from_numpy(padded).to("cuda:0")
        c.bind_device_frames(d.data_ptr(), n, h, w, stride=3 * w + stride_pad, frame_pitch=(3 * w + stride_pad) * h, keepalive=d)
    else:
        c.upload(frames)
    before = ws_launches()
    c.run(p, STAGE_ALL)
    c.sync()
    assert (ws_launches() - before > 0) == expect_ws
    assert c.check_guards()[0] == 0
    with ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(lambda f: oracle.detect_frame(f, p), frames))
    arm, offs = c.armours()
    for f, r in enumerate(refs):
        assert np.array_equal(c.binary(f), r["binary"]), f
        pts, co = c.contours(f)
        assert np.array_equal(co, r["offs"]) and np.array_equal(pts, r["pts"]), f
        assert arm[offs[f]:offs[f + 1]].tobytes() == r["armours"].tobytes(), f
    c.close()
    return refs


@pytest.mark.parametrize("morph", [MORPH_CLOSE, MORPH_DILATE, MORPH_NONE], ids=["close", "dilate", "none"])
@pytest.mark.parametrize("w,h,n", [(1280, 1024, 24), (1920, 1200, 6), (640, 480, 40), (1280, 1000, 9), (192, 70, 64), (2560, 96, 48),
                                    (4096, 40, 70), (64, 33, 160)])
def test_ws_kernel_equals_oracle_on_every_frame(oracle, w, h, n, morph):
    """widths of 1 .. 64 words per row (one and several passes per storer wavefront, with and without row masks), heights that end
    inside a strip and inside a storer wavefront's rows, strips at the image's top and bottom; sparse and noisy frames"""
    frames = make_frames(7000 + w + h, n, w, h)
    refs = run_and_check(oracle, frames, CAMP_BLUE, morph)
    if w >= 640 and h >= 400:
        assert sum(len(r["armours"]) for r in refs) > 0


@pytest.mark.parametrize("camp", [CAMP_RED, CAMP_GUIDELIGHT], ids=["red", "guidelight"])
def test_ws_kernel_other_camps_and_bounds(oracle, camp):
    frames = synth.batch(4321, 12, 1280, 1024, camp if camp == CAMP_RED else CAMP_BLUE, 1, threads=16)
    run_and_check(oracle, frames, camp, MORPH_CLOSE)
    run_and_check(oracle, frames, camp, MORPH_CLOSE, lb=1)
    run_and_check(oracle, frames, camp, MORPH_DILATE, lb=256)


def test_what_the_ws_kernel_does_not_take_falls_back(oracle):
    """lb <= 0 (everything passes), rows with padding between them, a launch with fewer strips than half the CUs (single frames are
    handed out as pieces): k_binary runs, the results are the same"""
    frames = synth.batch(1234, 10, 1280, 1024, CAMP_BLUE, 0, threads=16)
    run_and_check(oracle, frames, CAMP_BLUE, MORPH_CLOSE, lb=0, expect_ws=False)
    run_and_check(oracle, frames, CAMP_BLUE, MORPH_CLOSE, stride_pad=64, expect_ws=False)
    run_and_check(oracle, frames[:3], CAMP_BLUE, MORPH_CLOSE, expect_ws=False)
    run_and_check(oracle, frames, CAMP_BLUE, MORPH_CLOSE, expect_ws=True)


def test_ws_kernel_full_batch_equals_k_binary_bit_for_bit():
    """256 x 1280x1024 (C3) and 256 x 1920x1200 (C5's geometry): byte image and bit plane consumers (contours) of EVERY frame against
    k_binary's -- which the other tests of this suite hold against the oracle at full size"""
    for w, h in ((1280, 1024), (1920, 1200)):
        frames = synth.batch(55 + w, 256, w, h, CAMP_BLUE, 1, threads=16)
        c = Context(device=0, max_frames=256, max_width=w, max_height=h)
        c.upload(frames)
        p = default_params()
        got = []
        for shape in (0, 1):
            c.set_option(OPT_PIXEL_SHAPE, shape)
            before = ws_launches()
            c.run(p, STAGE_ALL)
            c.sync()
            assert (ws_launches() > before) == bool(shape)
            got.append(([c.binary(f).copy() for f in range(256)], [c.contours(f) for f in range(256)], c.armours()))
        for f in range(256):
            assert np.array_equal(got[0][0][f], got[1][0][f]), f
            assert np.array_equal(got[0][1][f][0], got[1][1][f][0]) and np.array_equal(got[0][1][f][1], got[1][1][f][1]), f
        assert got[0][2][0].tobytes() == got[1][2][0].tobytes() and len(got[0][2][0]) > 100
        assert c.check_guards()[0] == 0
        c.close()
