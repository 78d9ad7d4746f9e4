"""findContours beyond its LDS tables: the MID TIER (the cycle formulation with its tables in global memory) against the oracle.

cv::findContours has no bound (src/imgproc.cpp:71-72); round 2 held a frame of more than 4096 border visits / 1024 non-empty words /
512 contours / 32 junction pixels in no form but the sequential scanner.  The dense synthetic streams (rmcv_amd/csrc/synth.c,
variants 10..14: up to 2000 extra specks and 13 bright windows per frame) are made of exactly such frames."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from rmcv_amd import (CAMP_BLUE, FRAME_MID_PATH, FRAME_SLOW_PATH, MORPH_NONE, OPT_CONTOUR_TIER, OPT_DENSE_DEFER, OPT_SPARSE_WAVES, STAGE_ALL, Context,
                      default_params, synth)

pytestmark = pytest.mark.gpu


def compare_batch(c, frames, refs):
    arm, offs = c.armours()
    cnt = c.counts()
    for f in range(len(frames)):
        r = refs[f]
        assert np.array_equal(c.binary(f), r["binary"]), f
        pts, co = c.contours(f)
        assert np.array_equal(co, r["offs"]) and np.array_equal(pts, r["pts"]), f
        blobs, src = c.blobs(f)
        assert blobs.tobytes() == r["blobs"].tobytes(), f
        assert arm[offs[f]:offs[f + 1]].tobytes() == r["armours"].tobytes(), f
    return cnt["status"]


@pytest.mark.parametrize("level,waves,defer", [(1, 4, 0), (2, 8, 0), (3, 4, 0), (4, 8, 0), (4, 4, 0), (1, 4, 1), (4, 4, 1)])
def test_dense_streams_full_batch_every_stage(oracle, level, waves, defer):
    """256 x 1280x1024 frames of a dense stream through the whole path, every stage of every frame against the oracle; no frame
    may need the sequential scanner, and the levels whose frames exceed the LDS tables must have taken the mid tier"""
    n = 256
    frames = synth.batch(7000 * level, n, 1280, 1024, CAMP_BLUE, 10 + level, threads=16)
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024, max_contours=4096)
    c.set_option(OPT_SPARSE_WAVES, waves)
    c.set_option(OPT_DENSE_DEFER, defer)
    c.upload(frames)
    c.run(default_params(), STAGE_ALL)
    c.sync()
    with ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(lambda f: oracle.detect_frame(frames[f]), range(n)))
    st = compare_batch(c, frames, refs)
    assert not (st & 15).any() and not (st & FRAME_SLOW_PATH).any()
    mid = int(np.count_nonzero(st & FRAME_MID_PATH))
    over = sum(1 for r in refs if len(r["offs"]) - 1 > 512)      # more contours than the LDS tier keeps: certainly beyond it
    assert mid >= over and (level < 3 or mid == n) and (level > 1 or mid < n // 4)
    c.close()


@pytest.mark.parametrize("variant", [0, 1])
def test_mid_tier_forced_on_the_ordinary_streams(oracle, variant):
    """RMCV_OPT_CONTOUR_TIER = 2: every frame of the plain and the stress stream on the mid tier -- same results as on the LDS tier"""
    n = 64
    frames = synth.batch(31000, n, 1280, 1024, CAMP_BLUE, variant, threads=16)
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    c.set_option(OPT_CONTOUR_TIER, 2)
    c.upload(frames)
    c.run(default_params(), STAGE_ALL)
    c.sync()
    with ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(lambda f: oracle.detect_frame(frames[f]), range(n)))
    st = compare_batch(c, frames, refs)
    assert (st == FRAME_MID_PATH).all()
    c.close()


@pytest.mark.parametrize("waves,defer", [(8, 1), (4, 1), (4, 0)])
def test_one_dense_frame_does_not_change_its_neighbours(oracle, waves, defer):
    """a clean batch with ONE dense frame in the middle: that frame takes the mid tier, the others stay on the LDS tables -- with
    8 wavefronts per frame, and with 4 both ways: the dense frame left to the second launch (RMCV_OPT_DENSE_DEFER) or finished by
    the first (the default)"""
    n = 16
    frames = synth.batch(52000, n, 1280, 1024, CAMP_BLUE, 0, threads=16)
    frames[7] = synth.frame(52007, 1280, 1024, CAMP_BLUE, 14)
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    c.set_option(OPT_SPARSE_WAVES, waves)
    c.set_option(OPT_DENSE_DEFER, defer)
    c.upload(frames)
    c.run(default_params(), STAGE_ALL)
    c.sync()
    refs = [oracle.detect_frame(frames[f]) for f in range(n)]
    st = compare_batch(c, frames, refs)
    assert st[7] == FRAME_MID_PATH and not np.delete(st, 7).any()
    c.close()


def _check(c, canvas, oracle):
    img = np.zeros(canvas.shape + (3,), np.uint8)
    img[..., 0] = canvas
    pts, offs, binary = c.extract_color_csr(img, CAMP_BLUE, 80, MORPH_NONE)
    rp, ro = oracle.find_contours(canvas)
    assert np.array_equal(binary, canvas) and np.array_equal(offs, ro) and np.array_equal(pts, rp)
    return int(c.counts()["status"][0]), len(ro) - 1


def test_mid_tier_shapes(oracle):
    """the shapes the LDS tier's side tables were made for, in quantities only the mid tier holds: hundreds of 3- and 4-visit
    junction pixels, nesting five deep next to thousands of specks, noise at four densities, a border of 60 000 points"""
    c = Context(device=0, max_frames=1, max_width=2048, max_height=1536, max_contours=32768, max_points=1 << 19)
    rng = np.random.default_rng(8)
    b = np.zeros((600, 1200), np.uint8)
    for j in range(120):                                     # 120 'Y' junctions (centre visited three times) ...
        cx, cy = 10 + 9 * j, 100
        for i in range(1, 4):
            b[cy - i, cx - i] = b[cy - i, cx + i] = b[cy + i, cx] = 255
        b[cy, cx] = 255
    for j in range(100):                                     # ... and 100 'X' crossings of diagonals (centre visited four times)
        cx, cy = 12 + 11 * j, 300
        for i in range(1, 5):
            b[cy - i, cx - i] = b[cy - i, cx + i] = b[cy + i, cx - i] = b[cy + i, cx + i] = 255
        b[cy, cx] = 255
    st, n = _check(c, b, oracle)
    assert st == FRAME_MID_PATH and n == 220
    a = np.zeros((900, 1400), np.uint8)                      # rings nested five deep, siblings in every hole, specks everywhere
    for d in range(5):
        a[40 + 60 * d:860 - 60 * d, 40 + 60 * d:1360 - 60 * d] = 255
        a[60 + 60 * d:840 - 60 * d, 60 + 60 * d:1340 - 60 * d] = 0
    sp = rng.random(a.shape) < 0.004
    a[sp] = 255
    st, n = _check(c, a, oracle)
    assert st == FRAME_MID_PATH and n > 100
    for density in (0.002, 0.01, 0.03):                      # salt at three densities (the densest: ~20 000 contours)
        e = ((rng.random((768, 1024)) < density) * 255).astype(np.uint8)
        st, n = _check(c, e, oracle)
        assert st == FRAME_MID_PATH, density
    s = np.zeros((400, 600), np.uint8)                        # one serpentine border of ~55 000 points
    for k in range(0, 380, 8):
        s[k:k + 4, 10:590] = 255
        s[k + 4:k + 8, (10 if (k // 8) % 2 else 586):(14 if (k // 8) % 2 else 590)] = 255
    st, n = _check(c, s, oracle)
    assert st == FRAME_MID_PATH and n == 1
    z = ((rng.random((600, 800)) < 0.4) * 255).astype(np.uint8)      # beyond the mid tier too (> 131072 visits): the last resort, still exact
    st, n = _check(c, z, oracle)
    assert st & FRAME_SLOW_PATH
    c.close()


def test_mid_tier_fuzz_random_scenes(oracle):
    """the 300 random scenes of test_contours_fuzz_random_scenes (tools/fuzz_contours.py) with the mid tier forced"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_contours", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                             "tools", "fuzz_contours.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(2025)
    c = Context(device=0, max_frames=1, max_width=512, max_height=512, max_contours=8192, max_points=1 << 17)
    c.set_option(OPT_CONTOUR_TIER, 2)
    mid = 0
    for t in range(300):
        h, w = int(rng.integers(8, 300)), int(rng.integers(8, 400))
        st, _ = _check(c, fz.random_scene(rng, h, w), oracle)
        mid += st == FRAME_MID_PATH
    assert mid > 280                                          # (a few scenes nest deeper than the fixed point's 32 rounds: literal)
    c.close()


def test_without_the_mid_tiers_scratch_dense_frames_take_the_scanner(oracle, monkeypatch):
    """the mid tier's scratch block (4.5-5.7 MB per frame slot) is allocated when frames are bound, and if it cannot be had the tier
    is simply absent: frames beyond the LDS tables go to the sequential scanner -- same lists, RMCV_FRAME_SLOW_PATH instead of
    RMCV_FRAME_MID_PATH (context creation used to fail instead)"""
    n, w, h = 6, 1280, 1024
    fr = synth.batch(3300, n, w, h, CAMP_BLUE, 0, threads=8)
    fr[2] = synth.frame(3302, w, h, CAMP_BLUE, 14)
    refs = [oracle.detect_frame(f) for f in fr]
    for no_mid, bit in ((False, FRAME_MID_PATH), (True, FRAME_SLOW_PATH)):
        if no_mid:
            monkeypatch.setenv("RMCV_NO_MID", "1")
        c = Context(device=0, max_frames=n, max_width=w, max_height=h, max_contours=4096)
        c.upload(fr)
        c.run(default_params(), STAGE_ALL)
        c.sync()
        st = c.counts()["status"]
        assert st[2] & bit and not (st & 15).any() and np.count_nonzero(st & (FRAME_MID_PATH | FRAME_SLOW_PATH)) == 1, (no_mid, st)
        arm, offs = c.armours()
        for f in range(n):
            pts, co = c.contours(f)
            assert np.array_equal(co, refs[f]["offs"]) and np.array_equal(pts, refs[f]["pts"]), (no_mid, f)
            assert arm[offs[f]:offs[f + 1]].tobytes() == refs[f]["armours"].tobytes(), (no_mid, f)
        c.close()
