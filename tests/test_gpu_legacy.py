"""GPU parity of the legacy matcher (rm::MatchLightBlob / rm::FindLightBlobs / rm::LightBlobOverlap and
cv::minAreaRect), through the C-ABI, bit-for-bit against oracle/rmcv_oracle_legacy.c."""
import numpy as np
import pytest

from rmcv_amd import (CAMP_BLUE, CAMP_RED, MORPH_CLOSE, STAGE_ALL, STAGE_BINARY, STAGE_BLOBS, STAGE_CONTOURS, LegacyParams,
                      RmcvError, default_params, synth)
from test_oracle_legacy import P, bar_image, contours_of, random_blob_contours, rect_contour

pytestmark = pytest.mark.gpu


def rr(a):
    return tuple(float(a[k]) for k in ("cx", "cy", "w", "h", "angle"))


def test_min_area_rect_kats(ctx, oracle):
    cases = [rect_contour(10, 20, 30, 70), rect_contour(0, 0, 5, 5), rect_contour(3, 4, 3, 40), rect_contour(3, 4, 90, 4),
             P([(7, 9)]), P([(1, 1), (2, 1), (3, 1), (2, 1)]), P([(0, 0), (1, 1), (2, 2), (3, 3), (2, 2), (1, 1)]),
             P([(5, 5), (5, 5), (5, 5)]), P([(2, 2), (2, 3), (3, 3), (3, 2)])]
    for c in cases:
        got, want = ctx.min_area_rect(c), oracle.min_area_rect(c)
        assert got.tobytes() == want.tobytes(), (c.tolist(), rr(got), rr(want))


@pytest.mark.parametrize("seed", range(8))
def test_min_area_rect_random_blobs(ctx, oracle, seed):
    n = 0
    for c in random_blob_contours(400 + seed, n_shapes=8, size=192):
        got, want = ctx.min_area_rect(c), oracle.min_area_rect(c)
        assert got.tobytes() == want.tobytes(), (seed, len(c), rr(got), rr(want))
        n += 1
    assert n > 3


def test_min_area_rect_noise_contours(ctx, oracle):
    rng = np.random.default_rng(11)
    n = 0
    for t in range(12):
        img = (rng.random((72, 96)) < rng.uniform(0.25, 0.7)).astype(np.uint8) * 255
        for c in contours_of(img):
            got, want = ctx.min_area_rect(c), oracle.min_area_rect(c)
            assert got.tobytes() == want.tobytes(), (t, len(c), rr(got), rr(want))
            n += 1
    assert n > 100


def test_min_area_rect_wide_contour(oracle):
    """a border as wide as the frame: the hull's column table at its full size"""
    from rmcv_amd import Context
    c = Context(device=0, max_frames=1, max_width=1920, max_height=1200)
    img = np.zeros((1200, 1920), np.uint8)
    yy, xx = np.mgrid[0:1200, 0:1920]
    img[((xx - 960) / 959.0) ** 2 + ((yy - 600) / 500.0) ** 2 <= 1] = 255
    img[100:1100, 0] = 255
    cont = max(contours_of(img), key=len)
    assert cont["x"].min() == 0 and cont["x"].max() == 1919
    assert c.min_area_rect(cont).tobytes() == oracle.min_area_rect(cont).tobytes()


def test_min_area_rect_rejects_open_point_sets(ctx):
    with pytest.raises(RmcvError):
        ctx.min_area_rect(P([(0, 0), (10, 0), (10, 10)]))         # columns 1..9 hold no point: not a findContours border


def test_match_lightblob_branches(ctx, oracle):
    _, mask = bar_image(10, (255, 0, 0))
    c = contours_of(mask)[0]
    area = oracle.contour_area(c)
    for args in [(1.5, 80, 70, 10, 99999, True), (1.5, 80, 70, 10, 99999, False), (1.5, 80, 70, area + 1, 99999, True),
                 (1.5, 80, 70, area, area, True), (1.5, 80, 70, 10, area - 1, True), (20, 80, 70, 10, 99999, True),
                 (1.5, 2, 70, 10, 99999, False), (1.5, 80, 5, 10, 99999, True), (1.5, 80, 5, 10, 99999, False)]:
        ok_g, box_g = ctx.match_lightblob(c, *args)
        ok_o, box_o = oracle.match_lightblob(c, *args)
        assert ok_g == ok_o, args
        if ok_o:
            assert box_g.tobytes() == box_o.tobytes(), args
    assert ctx.match_lightblob(c[:5], 1.5, 80, 70, 0, 99999, True)[0] is False


@pytest.mark.parametrize("fit_ellipse", [True, False])
def test_find_lightblobs_synthetic_frames(ctx, oracle, fit_ellipse):
    """FindLightBlobs on the contours of the synthetic stream (both camps' bars are present in it)"""
    total = 0
    for k in range(4):
        frame = synth.frame(50 + k, 1280, 1024, CAMP_BLUE, k & 1)
        pts, offs, _ = ctx.extract_color_csr(frame, CAMP_RED if k == 3 else CAMP_BLUE, 80, MORPH_CLOSE)
        args = (1.5, 80, 70, 10, 99999)
        gb, gs, gx = ctx.find_lightblobs(pts, offs, *args, frame, fit_ellipse)
        ob, os_, ox = oracle.find_lightblobs(frame, pts, offs, *args, fit_ellipse)
        assert np.array_equal(gs, os_), k
        assert gx.tobytes() == ox.tobytes(), k
        assert gb.tobytes() == ob.tobytes(), k
        total += len(ob)
    assert total > 8


def test_find_lightblobs_camp_vote(ctx, oracle):
    img = np.zeros((200, 300, 3), np.uint8)
    mask = np.zeros((200, 300), np.uint8)
    for k, (x, colour) in enumerate([(40, (255, 0, 0)), (120, (0, 0, 255)), (200, (0, 255, 0)), (250, (90, 90, 90))]):
        img[30 + 30 * k:90 + 30 * k, x:x + 8] = colour
        mask[30 + 30 * k:90 + 30 * k, x:x + 8] = 255
    pts, offs = oracle.find_contours(mask)
    for fit in (True, False):
        gb, gs, _ = ctx.find_lightblobs(pts, offs, 1.5, 80, 70, 10, 99999, img, fit)
        ob, os_, _ = oracle.find_lightblobs(img, pts, offs, 1.5, 80, 70, 10, 99999, fit)
        assert gb.tobytes() == ob.tobytes() and np.array_equal(gs, os_)
        assert sorted(int(t) for t in gb["target"]) == [0, 0, 1, 2]


@pytest.mark.parametrize("fit_ellipse", [1, 0])
def test_batch_run_legacy(ctx, oracle, fit_ellipse):
    """batch path: extract_color -> FindLightBlobs -> filter_armours on the blobs of the enemy camp"""
    n = 6
    frames = synth.batch(300, n, 1280, 1024, CAMP_BLUE, 1)
    lp = LegacyParams(1.5, 80, 70, 10, 99999, fit_ellipse)
    p = default_params()
    ctx.upload(frames)
    ctx.run_legacy(lp, p, STAGE_ALL)
    ctx.sync()
    arm, aoffs = ctx.armours()
    assert not (ctx.counts()["status"] & ~16).any()
    na = 0
    for f in range(n):
        ref = oracle.detect_frame(frames[f], oracle.default_params())
        ob, _, _ = oracle.find_lightblobs(frames[f], ref["pts"], ref["offs"], 1.5, 80, 70, 10, 99999, fit_ellipse)
        gb, _ = ctx.blobs(f)
        assert gb.tobytes() == ob.tobytes(), f
        oa = oracle.filter_armours(ob, oracle.default_params())
        a = arm[aoffs[f]:aoffs[f + 1]]
        assert a.tobytes() == oa.tobytes(), f
        na += len(oa)
    assert na > 0
    # and the blobs-only stage mask
    ctx.run_legacy(lp, p, STAGE_BINARY | STAGE_CONTOURS | STAGE_BLOBS)
    ctx.sync()
    gb, _ = ctx.blobs(0)
    ref = oracle.detect_frame(frames[0], oracle.default_params())
    assert gb.tobytes() == oracle.find_lightblobs(frames[0], ref["pts"], ref["offs"], 1.5, 80, 70, 10, 99999, fit_ellipse)[0].tobytes()


def test_lightblob_overlap(ctx, oracle):
    rng = np.random.default_rng(3)
    blobs = np.zeros(12, oracle.LIGHTBLOB)
    for i in range(12):
        box = np.array((20 + 25 * i + rng.uniform(-5, 5), rng.uniform(40, 80), 6, rng.uniform(20, 60), rng.uniform(-10, 10)), oracle.RRECT)
        blobs[i] = oracle.make_lightblob(box, int(rng.integers(0, 2)))
    for left in range(-1, 12):
        for right in range(0, 13):
            want = oracle.lightblob_overlap(blobs, left, right)
            if want < 0:
                with pytest.raises(RmcvError):
                    ctx.lightblob_overlap(blobs, left, right)
            else:
                assert ctx.lightblob_overlap(blobs, left, right) == bool(want), (left, right)


def test_full_size_legacy_properties(oracle):
    """BASELINE batch size (256 x 1280x1024).  Size-independent properties: (1) with fitEllipse=true and the same bounds the
    legacy matcher accepts exactly the contours rm::filter_lightblobs calls positive (strict vs inclusive compares differ only
    for NaN) and boxes them identically -- only the camp is voted instead of given; (2) the minAreaRect box of every blob
    encloses its contour and is no larger than the axis-aligned bounding box; (3) spot frames equal the oracle bit for bit."""
    from rmcv_amd import Context
    n = 256
    frames = synth.batch(7000, n, 1280, 1024, CAMP_BLUE, 0)
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    c.upload(frames)
    p = default_params()
    c.run(p, STAGE_ALL)
    c.sync()
    base = [c.blobs(f) for f in range(n)]
    arm0, offs0 = c.armours()
    c.run_legacy(LegacyParams(1.5, 80, 70, 10, 99999, 1), p, STAGE_ALL)
    c.sync()
    arm1, offs1 = c.armours()
    blue = 0
    for f in range(n):
        b0, s0 = base[f]
        b1, s1 = c.blobs(f)
        assert np.array_equal(s0, s1), f
        for k in ("angle", "center", "vertices", "size"):
            assert b0[k].tobytes() == b1[k].tobytes(), (f, k)
        blue += int(np.count_nonzero(b1["target"] == CAMP_BLUE))
    assert blue > n                                             # the stream's enemy bars vote blue
    # armours of the legacy run = pairs among the blobs voted blue: a subset of the current API's armours per frame
    assert offs1[-1] <= offs0[-1] and offs1[-1] > 0
    c.run_legacy(LegacyParams(1.5, 80, 70, 10, 99999, 0), p, STAGE_ALL)
    c.sync()
    assert not (c.counts()["status"] & ~16).any()
    for f in range(0, n, 37):
        pts, offs = c.contours(f)
        blobs, src = c.blobs(f)
        ob, os_, ox = oracle.find_lightblobs(frames[f], pts, offs, 1.5, 80, 70, 10, 99999, False)
        assert np.array_equal(src, os_) and blobs.tobytes() == ob.tobytes(), f
        for b, s, box in zip(blobs, src, ox):
            cont = pts[offs[s]:offs[s + 1]]
            w, h, ang = float(box["w"]), float(box["h"]), np.deg2rad(float(box["angle"]))
            e = np.array([np.cos(ang), np.sin(ang)])
            d = np.stack([cont["x"], cont["y"]], 1).astype(float) - np.array([float(box["cx"]), float(box["cy"])])
            assert np.all(np.abs(d @ e) <= w / 2 + 1e-2) and np.all(np.abs(d @ np.array([-e[1], e[0]])) <= h / 2 + 1e-2)
            bw, bh = np.ptp(cont["x"]), np.ptp(cont["y"])
            assert w * h <= bw * bh + 1e-3 * max(1.0, bw * bh)
    c.close()


def test_find_lightblobs_contours_beyond_the_small_hull_tables(ctx, oracle):
    """the matcher's first pass holds hulls of <= 128 points over <= 256 columns (a disc of radius 127 has 84 hull points and
    fits); a 700-column ellipse and a 300-column bar take the second, full-size pass"""
    h, w = 700, 1280
    img = np.zeros((h, w, 3), np.uint8)
    mask = np.zeros((h, w), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    shapes = [((xx - 150) ** 2 + (yy - 150) ** 2 <= 127 ** 2, (255, 30, 0)),
              (((xx - 800) / 350.0) ** 2 + ((yy - 200) / 60.0) ** 2 <= 1, (0, 30, 255)),
              ((np.abs(xx - 400) <= 150) & (np.abs(yy - 500) <= 12), (20, 255, 20)),
              ((np.abs(xx - 900) <= 4) & (np.abs(yy - 500) <= 40), (255, 0, 0))]
    for m, colour in shapes:
        img[m] = colour
        mask[m] = 255
    pts, offs = oracle.find_contours(mask)
    gb, gs, gx = ctx.find_lightblobs(pts, offs, 0.5, 80, 180, 10, 1e9, img, False)
    ob, os_, ox = oracle.find_lightblobs(img, pts, offs, 0.5, 80, 180, 10, 1e9, False)
    assert len(ob) == 4 and np.array_equal(gs, os_)
    assert gx.tobytes() == ox.tobytes() and gb.tobytes() == ob.tobytes()
    widths = sorted(int(np.ptp(pts[offs[i]:offs[i + 1]]["x"])) + 1 for i in range(4))
    assert widths[0] <= 256 and widths[1] <= 256 and widths[2] > 256 and widths[3] > 256
