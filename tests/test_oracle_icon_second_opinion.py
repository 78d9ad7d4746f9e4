"""SURVEY 8f-1 / VERDICT r2 #8: a second opinion for the [OCV] fixed-point pieces of the icon path (warpAffine + resize on 8-bit
BGR, getAffineTransform), which so far had ONE restatement (oracle/rmcv_oracle.c, mirrored by k_classify).  tests/independent_icon.py
restates them a second time, table-driven and vectorised, from the published structure; here the two are compared bit for bit on
more than a thousand regions.  Neither is pinned against real OpenCV (absent here); agreement removes slips of one author's one
restatement, not shared misconceptions."""
import numpy as np

import independent_icon as ind
import oracle_lib as O


def scene(rng, h, w):
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[..., 1] = (np.arange(w)[None, :] * 255 // max(1, w - 1)).astype(np.uint8)      # a ramp: interpolation errors cannot hide in noise
    yy, xx = np.mgrid[0:h, 0:w]
    img[..., 2] = ((xx // 7 + yy // 5) % 2 * 200 + 20).astype(np.uint8)                # and a checkerboard: every tap matters
    return img


def random_icon(rng, h, w):
    kind = rng.integers(0, 6)
    cx, cy = rng.uniform(0, w), rng.uniform(0, h)
    if kind == 0:                                             # exact 40 x 40 box: the 2:1 area branch of resize
        x0, y0 = int(rng.integers(0, w - 41)), int(rng.integers(0, h - 41))
        return np.array([[x0, y0 + 39], [x0, y0], [x0 + 39, y0], [x0 + 39, y0 + 39]], np.float32)
    if kind == 1:                                             # tiny: 1..4 px boxes (every column clamped in resize)
        s = rng.uniform(0.3, 3.5)
        q = np.array([[-s, s], [-s, -s], [s, -s], [s, s]]) * 0.5 + (cx, cy)
        return q.astype(np.float32)
    sx, sy = rng.uniform(4, 160), rng.uniform(4, 160)
    ang = rng.uniform(-0.6, 0.6)
    base = np.array([[-sx, sy], [-sx, -sy], [sx, -sy], [sx, sy]]) * 0.5        # [0] bottom-left, [1] top-left, [2] top-right, [3] bottom-right
    if kind >= 4:
        base += rng.normal(0, 0.08 * min(sx, sy), base.shape)                  # a sheared / perspective-looking quad
    r = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    q = base @ r.T + (cx, cy)
    if kind == 3:
        q += rng.uniform(-0.4, 0.4) * np.array([w, h])                          # partly (or wholly) outside: the clamp of imgproc.cpp:11-15
    return q.astype(np.float32)


def test_warp_resize_second_opinion_on_1200_regions():
    rng = np.random.default_rng(20241008)
    n = same = clamped = area = 0
    for s in range(24):
        h, w = int(rng.integers(60, 400)), int(rng.integers(60, 500))
        img = scene(rng, h, w)
        for _ in range(50):
            icon = random_icon(rng, h, w)
            a, ia, ra = O.affine_correction(img, icon)
            b, ib, rb = ind.affine_correction(img, icon)
            assert ra == rb and np.array_equal(ia, ib), (s, icon.tolist())
            assert np.array_equal(a, b), (s, icon.tolist(), int(np.abs(a.astype(int) - b.astype(int)).max()))
            n += 1
            same += ra == 0
            clamped += not np.array_equal(ia, icon)
            area += (ia[:, 0].max() - ia[:, 0].min() == 39) and (ia[:, 1].max() - ia[:, 1].min() == 39)
    assert n == 1200 and same > 1100 and clamped > 100 and area > 100


def test_weight_table_is_the_closed_form_except_where_shorts_saturate():
    """the table built OpenCV's way equals 32 (32 - fx)(32 - fy) ... everywhere but at (0, 0), where 2^15 does not fit a short:
    32767 there, and the missing 1 goes to the LAST tap -- what the oracle's closed form does with `w11 += 1`"""
    t = ind.bilinear_table()
    for fy in range(32):
        for fx in range(32):
            cf = [[(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32], [(32 - fx) * fy * 32, fx * fy * 32]]
            if fx == 0 and fy == 0:
                assert t[0, 0].tolist() == [[32767, 0], [0, 1]]
            else:
                assert t[fy, fx].tolist() == cf, (fx, fy)


def test_synthetic_stream_icons_second_opinion():
    """the icons the benchmark's own C5 frames produce (every armour of 24 frames at 1920x1200)"""
    from rmcv_amd import synth
    O.set_math_mode(0)
    n = 0
    for i in range(24):
        fr = synth.frame(120000 + i, 1920, 1200)
        for a in O.detect_frame(fr)["armours"]:
            x, ic, rc = O.affine_correction(fr, a["icon"])
            y, ic2, rc2 = ind.affine_correction(fr, a["icon"])
            assert rc == rc2 and np.array_equal(ic, ic2) and np.array_equal(x, y)
            n += 1
    assert n > 40
