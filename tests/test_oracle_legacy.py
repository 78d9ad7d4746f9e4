"""CPU tests of the legacy-matcher oracle: convex hull, minAreaRect, MatchLightBlob, FindLightBlobs,
LightBlobOverlap.  The reference has no tests for these; the known answers below are this build's own
(hand-derived or computed by an independent method: scipy's Qhull, a brute-force minimum over hull edges in float64)."""
import numpy as np
import pytest

import oracle_lib as O


def P(lst):
    a = np.zeros(len(lst), O.POINT)
    for i, (x, y) in enumerate(lst):
        a[i] = (x, y)
    return a


def rect_contour(x0, y0, x1, y1):
    """findContours order of a filled rectangle: down the left side, along the bottom, up the right side, back along the top"""
    pts = [(x0, y) for y in range(y0, y1 + 1)]
    pts += [(x, y1) for x in range(x0 + 1, x1 + 1)]
    pts += [(x1, y) for y in range(y1 - 1, y0 - 1, -1)]
    pts += [(x, y0) for x in range(x1 - 1, x0, -1)]
    return P(pts)


def contours_of(img):
    pts, offs = O.find_contours(img)
    return [pts[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]


def random_blob_contours(seed, n_shapes=6, size=160):
    rng = np.random.default_rng(seed)
    img = np.zeros((size, size), np.uint8)
    yy, xx = np.mgrid[0:size, 0:size]
    for _ in range(n_shapes):
        cx, cy = rng.integers(20, size - 20, 2)
        a, b = rng.integers(3, 18), rng.integers(3, 30)
        th = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        if rng.integers(0, 2):
            img[(u / a) ** 2 + (v / b) ** 2 <= 1] = 255
        else:
            img[(np.abs(u) <= a) & (np.abs(v) <= b)] = 255
    img[rng.random(img.shape) < 0.01] = 255      # spurs and specks: duplicate points in the contours
    return contours_of(img)


# ------------------------------------------------------------------------------------------------ convex hull
def test_hull_rectangle_vertices_and_orientation():
    c = rect_contour(10, 20, 30, 70)
    h = O.convex_hull(c)
    hp = [(int(c[i]["x"]), int(c[i]["y"])) for i in h]
    assert sorted(hp) == sorted([(10, 20), (10, 70), (30, 70), (30, 20)])
    # clockwise=false: counter-clockwise in OpenCV's documented convention (x right, y UP) = positive shoelace sum on raw coordinates
    x = np.array([p[0] for p in hp], float)
    y = np.array([p[1] for p in hp], float)
    assert np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y) > 0
    # the contour's own index order is already monotone along the hull, so the cyclic shift starts it at the lowest index
    assert list(h) == sorted(h) or list(h) == sorted(h, reverse=True)


def test_hull_degenerate_cases():
    assert list(O.convex_hull(P([(5, 5)]))) == [0]
    assert list(O.convex_hull(P([(5, 5), (5, 5), (5, 5)]))) == [0]
    line = P([(1, 1), (2, 1), (3, 1), (2, 1)])               # findContours of a 1x3 line
    h = O.convex_hull(line)
    assert sorted((int(line[i]["x"]), int(line[i]["y"])) for i in h) == [(1, 1), (3, 1)]
    diag = P([(0, 0), (1, 1), (2, 2), (3, 3), (2, 2), (1, 1)])
    h = O.convex_hull(diag)
    assert sorted((int(diag[i]["x"]), int(diag[i]["y"])) for i in h) == [(0, 0), (3, 3)]


@pytest.mark.parametrize("seed", range(6))
def test_hull_matches_qhull_vertex_set(seed):
    from scipy.spatial import ConvexHull, QhullError
    for c in random_blob_contours(seed):
        if len(c) < 3:
            continue
        xy = np.stack([c["x"], c["y"]], 1).astype(float)
        try:
            q = ConvexHull(xy)
        except QhullError:
            continue                                          # collinear input
        want = {tuple(map(int, xy[v])) for v in q.vertices}
        h = O.convex_hull(c)
        got = [(int(c[i]["x"]), int(c[i]["y"])) for i in h]
        assert len(set(got)) == len(got), "a hull vertex repeats"
        assert set(got) == want


@pytest.mark.parametrize("seed", range(10))
def test_hull_column_pruning_is_equivalent(seed):
    """the HIP kernel scans four entries per column instead of all sorted points: same indices, same order"""
    n = 0
    for c in random_blob_contours(100 + seed, n_shapes=8):
        a, b = O.convex_hull(c), O.convex_hull(c, pruned=True)
        assert b is not None and list(a) == list(b), (len(c), list(a), list(b))
        n += 1
    assert n > 3


# ------------------------------------------------------------------------------------------------ minAreaRect
def brute_min_area(c):
    h = O.convex_hull(c)
    hp = np.stack([c["x"][h], c["y"][h]], 1).astype(float)
    best = None
    for i in range(len(hp)):
        e = hp[(i + 1) % len(hp)] - hp[i]
        e /= np.hypot(*e)
        nrm = np.array([-e[1], e[0]])
        u, v = hp @ e, hp @ nrm
        area = (u.max() - u.min()) * (v.max() - v.min())
        if best is None or area < best:
            best = area
    return best


def test_min_area_rect_axis_aligned_rectangle():
    r = O.min_area_rect(rect_contour(10, 20, 30, 70))
    assert (float(r["cx"]), float(r["cy"])) == (20.0, 45.0)
    assert sorted([float(r["w"]), float(r["h"])]) == [20.0, 50.0]
    assert float(r["angle"]) in (0.0, 90.0, -90.0, 180.0)
    # size is tied to the angle: rotating (w, h) by the angle gives the box again
    if float(r["angle"]) in (0.0, 180.0):
        assert (float(r["w"]), float(r["h"])) == (20.0, 50.0)
    else:
        assert (float(r["w"]), float(r["h"])) == (50.0, 20.0)


def test_min_area_rect_degenerate():
    r = O.min_area_rect(P([(7, 9)]))
    assert tuple(float(r[k]) for k in ("cx", "cy", "w", "h", "angle")) == (7.0, 9.0, 0.0, 0.0, 0.0)
    r = O.min_area_rect(P([(1, 1), (2, 1), (3, 1), (2, 1)]))
    assert (float(r["cx"]), float(r["cy"]), float(r["h"])) == (2.0, 1.0, 0.0) and float(r["w"]) == 2.0
    assert float(r["angle"]) in (0.0, 180.0)


@pytest.mark.parametrize("seed", range(6))
def test_min_area_rect_is_minimal_and_encloses(seed):
    for c in random_blob_contours(200 + seed):
        h = O.convex_hull(c)
        if len(h) < 3:
            continue
        r = O.min_area_rect(c)
        w, hh, ang = float(r["w"]), float(r["h"]), np.deg2rad(float(r["angle"]))
        assert abs(w * hh - brute_min_area(c)) <= 1e-3 * max(1.0, w * hh)
        e = np.array([np.cos(ang), np.sin(ang)])
        nrm = np.array([-e[1], e[0]])
        d = np.stack([c["x"], c["y"]], 1).astype(float) - np.array([float(r["cx"]), float(r["cy"])])
        assert np.all(np.abs(d @ e) <= w / 2 + 1e-2) and np.all(np.abs(d @ nrm) <= hh / 2 + 1e-2)


def test_min_area_rect_math_modes_agree():
    for c in random_blob_contours(300):
        O.set_math_mode(1)
        a = O.min_area_rect(c)
        O.set_math_mode(0)
        b = O.min_area_rect(c)
        for k in ("cx", "cy", "w", "h"):
            assert a[k] == b[k]
        assert abs(float(a["angle"]) - float(b["angle"])) <= 1e-4


# ------------------------------------------------------------------------------------------------ matcher
def bar_image(angle_deg, colour, size=120, half_w=4, half_h=25):
    img = np.zeros((size, size, 3), np.uint8)
    yy, xx = np.mgrid[0:size, 0:size]
    th = np.deg2rad(angle_deg)
    u = (xx - size / 2) * np.cos(th) + (yy - size / 2) * np.sin(th)
    v = -(xx - size / 2) * np.sin(th) + (yy - size / 2) * np.cos(th)
    m = (np.abs(u) <= half_w) & (np.abs(v) <= half_h)
    img[m] = colour
    return img, (m * 255).astype(np.uint8)


def test_match_lightblob_branches():
    _, mask = bar_image(10, (255, 0, 0))
    c = contours_of(mask)[0]
    area = O.contour_area(c)
    ok, box = O.match_lightblob(c, 1.5, 80, 70, 10, 99999, True)
    assert ok
    e, _ = O.fit_ellipse_direct(c)
    assert box == e
    ok, box = O.match_lightblob(c, 1.5, 80, 70, 10, 99999, False)
    assert ok and box == O.min_area_rect(c)
    assert not O.match_lightblob(c[:5], 1.5, 80, 70, 0, 99999, True)[0]          # fewer than 6 points
    assert not O.match_lightblob(c, 1.5, 80, 70, area + 1, 99999, True)[0]       # area < minArea
    assert O.match_lightblob(c, 1.5, 80, 70, area, area, True)[0]                # bounds are strict compares: equal passes
    assert not O.match_lightblob(c, 1.5, 80, 70, 10, area - 1, True)[0]          # area > maxArea
    assert not O.match_lightblob(c, 20, 80, 70, 10, 99999, True)[0]              # ratio < minRatio
    assert not O.match_lightblob(c, 1.5, 2, 70, 10, 99999, True)[0]              # ratio > maxRatio
    assert not O.match_lightblob(c, 1.5, 80, 5, 10, 99999, True)[0]              # tilt 10 deg > 5
    assert not O.match_lightblob(c, 1.5, 80, 5, 10, 99999, False)[0]             # the tilt always comes from the ellipse


@pytest.mark.parametrize("colour,camp", [((255, 40, 10), O.CAMP_BLUE), ((10, 40, 255), O.CAMP_RED),
                                          ((40, 255, 10), O.CAMP_GUIDELIGHT), ((200, 200, 200), O.CAMP_RED)])
def test_find_lightblobs_camp_from_mean(colour, camp):
    img, mask = bar_image(-8, colour)
    pts, offs = O.find_contours(mask)
    for fit in (True, False):
        blobs, src, boxes = O.find_lightblobs(img, pts, offs, 1.5, 80, 70, 10, 99999, fit)
        assert len(blobs) == 1 and src[0] == 0 and int(blobs[0]["target"]) == camp
        assert blobs[0] == O.make_lightblob(boxes[0], camp)


def test_find_lightblobs_order_and_rejects():
    img = np.zeros((200, 300, 3), np.uint8)
    mask = np.zeros((200, 300), np.uint8)
    for k, (x, colour) in enumerate([(40, (255, 0, 0)), (120, (0, 0, 255)), (200, (0, 255, 0))]):
        img[30 + 40 * k:90 + 40 * k, x:x + 8] = colour
        mask[30 + 40 * k:90 + 40 * k, x:x + 8] = 255
    img[10:14, 260:264] = (255, 0, 0)                       # small square: ratio 1 -> rejected
    mask[10:14, 260:264] = 255
    pts, offs = O.find_contours(mask)
    blobs, src, _ = O.find_lightblobs(img, pts, offs, 1.5, 80, 70, 10, 99999, True)
    # findContours returns the lowest component first; FindLightBlobs keeps that order
    assert [int(b["target"]) for b in blobs] == [O.CAMP_GUIDELIGHT, O.CAMP_RED, O.CAMP_BLUE]
    assert list(src) == sorted(src)


def test_lightblob_overlap():
    def blob(cx, cy, camp, half_h=20):
        return O.make_lightblob(np.array((cx, cy, 6, 2 * half_h, 0), O.RRECT), camp)
    blobs = np.array([blob(10, 50, 1), blob(50, 50, 1), blob(90, 50, 1)], O.LIGHTBLOB)
    assert O.lightblob_overlap(blobs, 0, 2) == 1
    assert O.lightblob_overlap(blobs, 0, 1) == 0            # rightIndex - leftIndex < 2
    assert O.lightblob_overlap(blobs, -1, 2) == 0
    assert O.lightblob_overlap(blobs, 0, 3) == -1           # the reference reads past the end here
    blobs2 = blobs.copy()
    blobs2[1]["target"] = 0                                 # the middle one belongs to the other camp
    assert O.lightblob_overlap(blobs2, 0, 2) == 0
    blobs3 = np.array([blob(10, 50, 1), blob(50, 120, 1), blob(90, 50, 1)], O.LIGHTBLOB)   # middle one far below
    assert O.lightblob_overlap(blobs3, 0, 2) == 0
    blobs4 = np.array([blob(10, 50, 1), blob(50, 50, 1), blob(90, 50, 0)], O.LIGHTBLOB)    # ends differ in camp
    assert O.lightblob_overlap(blobs4, 0, 2) == 0
