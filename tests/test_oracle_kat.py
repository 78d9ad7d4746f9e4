"""Known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors, so these KATs are THE BUILD'S OWN,
hand-derived from the published semantics of the OpenCV calls the reference makes (SURVEY.md 8c,
Appendix A) -- not fixtures of the reference.  They are what stands between the oracle and "unpinned".
"""
import json
import math
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def img_of(h, w, pix):
    a = np.zeros((h, w), np.uint8)
    for x, y in pix:
        a[y, x] = 255
    return a


def contours_of(binary):
    return O.contours_as_lists(*O.find_contours(binary))


# ---------------------------------------------------------------- KAT-1 threshold (imgproc.cpp:52-65)
def test_kat1_threshold():
    px = np.zeros((1, 6, 3), np.uint8)
    px[0, 0] = (200, 0, 120)   # B-R = 80  -> 255 at lb 80
    px[0, 1] = (199, 0, 120)   # 79        -> 0
    px[0, 2] = (10, 0, 200)    # saturates to 0
    px[0, 3] = (255, 255, 0)   # 255
    px[0, 4] = (120, 0, 200)   # red pixel
    px[0, 5] = (0, 200, 100)   # green - red = 100
    assert O.extract_binary(px, O.CAMP_BLUE, 80, O.MORPH_NONE).tolist() == [[255, 0, 0, 255, 0, 0]]
    assert O.extract_binary(px, O.CAMP_RED, 80, O.MORPH_NONE).tolist() == [[0, 0, 255, 0, 255, 255]]
    assert O.extract_binary(px, O.CAMP_NEUTRAL, 80, O.MORPH_NONE).tolist() == [[0, 0, 255, 0, 255, 255]]  # behaves like RED
    assert O.extract_binary(px, O.CAMP_GUIDELIGHT, 100, O.MORPH_NONE).tolist() == [[0, 0, 0, 255, 0, 255]]
    assert O.extract_binary(px, O.CAMP_BLUE, 0, O.MORPH_NONE).min() == 255      # gray >= 0 always
    assert O.extract_binary(px, O.CAMP_BLUE, 256, O.MORPH_NONE).max() == 0      # gray <= 255 < 256


# ---------------------------------------------------------------- KAT-2 close (imgproc.cpp:68-69)
def test_kat2_close():
    a = np.zeros((9, 9), np.uint8)
    a[2:7, 2:7] = 255
    a[4, 4] = 0                                    # 1-px hole inside a 5x5 block is filled
    c = O.erode3x3(O.dilate3x3(a))
    exp = np.zeros((9, 9), np.uint8)
    exp[2:7, 2:7] = 255
    assert np.array_equal(c, exp)
    s = img_of(7, 7, [(3, 3)])                     # a single pixel survives
    assert np.array_equal(O.erode3x3(O.dilate3x3(s)), s)
    k = img_of(5, 5, [(0, 0), (4, 4), (4, 0)])     # corner pixels survive: the border never wins
    assert np.array_equal(O.erode3x3(O.dilate3x3(k)), k)
    g = img_of(5, 9, [(2, 2), (4, 2)])             # a 1-px gap between two pixels is bridged
    assert O.erode3x3(O.dilate3x3(g))[2].tolist() == [0, 0, 255, 255, 255, 0, 0, 0, 0]
    d = O.dilate3x3(img_of(5, 5, [(0, 0)]))        # dilate at the corner: 2x2
    assert int(d.sum()) == 4 * 255 and d[1, 1] == 255
    full = np.full((4, 6), 255, np.uint8)          # erode of a full image stays full (border = +inf)
    assert np.array_equal(O.erode3x3(full), full)


# ---------------------------------------------------------------- KAT-3 findContours (imgproc.cpp:71-72)
def test_kat3_contours_order_and_points():
    assert contours_of(img_of(5, 5, [(2, 3)])) == [[(2, 3)]]
    a = np.zeros((8, 8), np.uint8)
    a[1:3, 1:3] = 255
    assert contours_of(a) == [[(1, 1), (1, 2), (2, 2), (2, 1)]]
    a = np.zeros((5, 6), np.uint8)
    a[1, 1:4] = 255
    assert contours_of(a) == [[(1, 1), (2, 1), (3, 1), (2, 1)]]      # spur pixels repeat
    a = np.zeros((10, 10), np.uint8)
    a[2:6, 3:8] = 255                                               # down the left, right along the bottom, up, back
    assert contours_of(a) == [[(3, 2), (3, 3), (3, 4), (3, 5), (4, 5), (5, 5), (6, 5), (7, 5), (7, 4), (7, 3), (7, 2),
                               (6, 2), (5, 2), (4, 2)]]
    two = img_of(10, 10, [(1, 1), (5, 7)])
    assert contours_of(two) == [[(5, 7)], [(1, 1)]]                 # the later (lower) one first
    v = np.zeros((6, 4), np.uint8)
    v[1:5, 1] = 255                                                 # vertical 1-px line: down then back up
    assert contours_of(v) == [[(1, 1), (1, 2), (1, 3), (1, 4), (1, 3), (1, 2)]]
    dg = img_of(5, 5, [(1, 1), (2, 2), (3, 3)])                     # 8-connected diagonal = one contour
    assert contours_of(dg) == [[(1, 1), (2, 2), (3, 3), (2, 2)]]
    edge = np.full((3, 3), 255, np.uint8)                           # touching the frame: traced in full
    assert contours_of(edge) == [[(0, 0), (0, 1), (0, 2), (1, 2), (2, 2), (2, 1), (2, 0), (1, 0)]]
    assert contours_of(np.zeros((4, 4), np.uint8)) == []


def test_kat3_external_rule():
    ring = np.zeros((12, 12), np.uint8)
    ring[1:10, 1:10] = 255
    ring[3:8, 3:8] = 0
    ring[5, 5] = 255                                                # blob inside the hole of a ring: dropped
    c = contours_of(ring)
    assert len(c) == 1 and c[0][0] == (1, 1)
    u = np.zeros((12, 12), np.uint8)
    u[1:10, 1:10] = 255
    u[1:8, 3:8] = 0
    u[4, 5] = 255                                                   # blob inside a "U" cavity: kept
    c = contours_of(u)
    assert len(c) == 2 and c[0] == [(5, 4)]
    thin = np.zeros((9, 9), np.uint8)                               # 1-px-thick ring: the hole is not reported
    thin[1, 1:8] = thin[7, 1:8] = 255
    thin[1:8, 1] = thin[1:8, 7] = 255
    assert len(contours_of(thin)) == 1


# ---------------------------------------------------------------- KAT-4 contourArea (objdetect.cpp:64)
def test_kat4_area():
    a = np.zeros((12, 14), np.uint8)
    a[2:9, 3:11] = 255                                              # x 3..10, y 2..8
    pts, offs = O.find_contours(a)
    assert O.contour_area(pts) == (10 - 3) * (8 - 2)
    line = np.zeros((5, 12), np.uint8)
    line[2, 1:10] = 255
    pts, _ = O.find_contours(line)
    assert O.contour_area(pts) == 0.0


# ---------------------------------------------------------------- KAT-5 fitEllipseDirect (objdetect.cpp:68)
def ellipse_points(cx, cy, A, B, th, n):
    P = np.zeros(n, O.POINT)
    for k in range(n):
        t = 2 * math.pi * k / n
        P["x"][k] = round(cx + A * math.cos(t) * math.cos(th) - B * math.sin(t) * math.sin(th))
        P["y"][k] = round(cy + A * math.cos(t) * math.sin(th) + B * math.sin(t) * math.cos(th))
    return P


@pytest.mark.parametrize("A,B,n,path,tol", [(400, 150, 720, 1, 2e-3), (36, 12, 40, 0, 3e-2)])
@pytest.mark.parametrize("deg", [0, 20, 60, 90, 135, 170])
def test_kat5_ellipse(A, B, n, path, tol, deg):
    r, used = O.fit_ellipse_direct(ellipse_points(700, 600, A, B, math.radians(deg), n))
    assert used == path          # small contours take the direct solution, large ones the general fit
    assert abs(r["cx"] - 700) < tol * A and abs(r["cy"] - 600) < tol * A
    assert r["w"] <= r["h"]
    assert abs(r["h"] - 2 * A) < tol * 2 * A and abs(r["w"] - 2 * B) < 3 * tol * 2 * B
    # the long (height) axis lies along deg: RotatedRect angle = deg + 90 (mod 180)
    d = abs((float(r["angle"]) - (deg + 90)) % 180)
    assert min(d, 180 - d) < 1.0
    if path == 0:
        assert 0 <= r["angle"] < 180


def test_kat5_translation():
    p = ellipse_points(300, 300, 36, 12, 0.4, 40)
    q = p.copy()
    q["x"] += 100
    q["y"] += 57
    a, _ = O.fit_ellipse_direct(p)
    b, _ = O.fit_ellipse_direct(q)
    assert abs((b["cx"] - a["cx"]) - 100) < 1e-3 and abs((b["cy"] - a["cy"]) - 57) < 1e-3
    assert abs(b["w"] - a["w"]) < 1e-3 and abs(b["h"] - a["h"]) < 1e-3 and abs(b["angle"] - a["angle"]) < 1e-3


# ---------------------------------------------------------------- KAT-6 lightblob / armour (core.cpp:9-49)
def rr(cx, cy, w, h, angle):
    r = np.zeros(1, O.RRECT)[0]
    r["cx"], r["cy"], r["w"], r["h"], r["angle"] = cx, cy, w, h, angle
    return r


def test_kat6_lightblob_armour():
    L = O.make_lightblob(rr(100, 100, 10, 40, 0), O.CAMP_BLUE)
    R = O.make_lightblob(rr(200, 100, 10, 40, 0), O.CAMP_BLUE)
    assert L["angle"] == 90 and L["target"] == O.CAMP_BLUE
    assert L["vertices"].tolist() == [[95, 120], [95, 80], [105, 80], [105, 120]]   # left-down, left-up, right-up, right-down
    assert L["size"].tolist() == [10, 40]
    assert O.make_lightblob(rr(0, 0, 10, 40, 120), 0)["angle"] == 30                 # >90 -> -90
    for a, b in ((L, R), (R, L)):                                                   # sorted left-to-right inside the ctor
        arm = O.make_armour(a, b)
        assert arm["vertices"].tolist() == [[130, 80], [130, 120], [170, 120], [170, 80]]  # square of side 40 between inner edges
        assert arm["icon"].tolist() == [[105, 140], [105, 60], [195, 60], [195, 140]]
        assert arm["bbox"].tolist() == [105, 60, 91, 81]


# ---------------------------------------------------------------- KAT-7 pairing (objdetect.cpp:114-166)
def blob(cx, cy, h, angle_rr=0.0, camp=O.CAMP_BLUE):
    return O.make_lightblob(rr(cx, cy, h / 4.0, h, angle_rr), camp)


def n_armours(blobs, **kw):
    return len(O.filter_armours(np.array(blobs, O.LIGHTBLOB), O.default_params(**kw)))


def test_kat7_pairing_rejects():
    a, b = blob(100, 100, 40), blob(200, 100, 40)
    assert n_armours([a, b]) == 1
    assert n_armours([a]) == 0                                           # :120
    assert n_armours([a, blob(200, 100, 40, camp=O.CAMP_RED)]) == 0      # :124-129 target filter
    assert n_armours([a, blob(200, 100, 40, 20.0)]) == 0                 # :131 angle difference 20 > 12
    assert n_armours([a, blob(200, 100, 40, 20.0)], angle_diff_max=25.0, shear_max=90.0) == 1
    assert n_armours([blob(100, 100, 40), blob(130, 118, 40)]) == 0      # :144 shear: centre line tilted ~31 deg > 22
    assert n_armours([blob(100, 100, 40), blob(130, 118, 40)], shear_max=40.0) == 1
    assert n_armours([a, blob(200, 100, 12)]) == 0                       # :149 12/40 < 0.4
    assert n_armours([a, blob(200, 100, 12)], length_ratio_max=0.2) == 1
    assert n_armours([blob(100, 100, 20), blob(160, 122, 20)], shear_max=90.0) == 0   # :153 |dy| 22 > 20
    assert n_armours([a, blob(265, 100, 40)]) == 0                       # :157 |dx| 165 > 160
    assert n_armours([a, blob(255, 100, 40)]) == 1
    three = O.filter_armours(np.array([a, blob(170, 100, 40), blob(240, 100, 40)], O.LIGHTBLOB))
    assert [(int(x["blob_i"]), int(x["blob_j"])) for x in three] == [(0, 1), (0, 2), (1, 2)]       # (i, j) lexicographic


def test_filter_lightblobs_gates():
    canvas = np.zeros((200, 300), np.uint8)
    canvas[20:100, 30:42] = 255        # upright bar: positive
    canvas[150:156, 50:130] = 255      # lying bar: tilt test fails -> negative
    canvas[120:123, 200:203] = 255     # 3x3: area 4 < 10 -> skipped entirely
    canvas[10, 250] = 255              # single pixel: < 6 points -> skipped
    pts, offs = O.find_contours(canvas)
    blobs, src, neg = O.filter_lightblobs(pts, offs)
    assert len(offs) - 1 == 4 and len(blobs) == 1 and len(neg) == 1
    starts = [tuple(pts[offs[i]]) for i in range(4)]
    assert starts[src[0]] == (30, 20) and starts[neg[0]] == (50, 150)
    assert abs(blobs[0]["angle"] - 90) < 1 and blobs[0]["size"][0] < blobs[0]["size"][1]


# ---------------------------------------------------------------- committed golden vectors
def test_golden_vectors():
    g = json.load(open(os.path.join(GOLD, "synthetic_armours.json")))
    from rmcv_amd import synth
    for rec in g["frames"]:
        f = synth.frame(rec["index"], g["width"], g["height"], g["camp"], rec["variant"])
        assert "%016x" % synth.checksum(f) == rec["frame_fnv1a"]
        r = O.detect_frame(f)
        assert len(r["offs"]) - 1 == rec["n_contours"] and len(r["pts"]) == rec["n_points"]
        assert len(r["blobs"]) == rec["n_blobs"]
        got = [[float.hex(float(v)) for v in a["vertices"].reshape(-1)] for a in r["armours"]]
        assert got == rec["armour_vertices_hex"]


def test_golden_vectors_next_rows():
    """identities, legacy light blobs and poses of the SURVEY 8f rows on the frozen frames (tests/golden/make_golden.py)"""
    import numpy as np
    g = json.load(open(os.path.join(GOLD, "synthetic_next_rows.json")))
    from rmcv_amd import synth
    svm = synth.svm_weights()
    for rec in g["frames"]:
        f = synth.frame(rec["index"], g["width"], g["height"], g["camp"], rec["variant"])
        r = O.detect_frame(f)
        ident, _, icons = O.classify_armours(f, r["armours"], svm)
        assert [int(v) for v in ident] == rec["identities"]
        assert [int(ic.astype(np.int64).sum()) for ic in icons] == rec["icon_sums"]
        lb, src, boxes = O.find_lightblobs(f, r["pts"], r["offs"], 1.5, 80, 70, 10, 99999, False)
        assert [int(v) for v in src] == rec["legacy_src"] and [int(b["target"]) for b in lb] == rec["legacy_targets"]
        assert [[float.hex(float(b[k])) for k in ("cx", "cy", "w", "h", "angle")] for b in boxes] == rec["legacy_boxes_hex"]
        rv, tv, pos = O.locate_armours(r["armours"])
        assert [[float.hex(float(v)) for v in t] for t in tv] == rec["tvec_hex"]
        assert [[float.hex(float(v)) for v in q] for q in pos] == rec["position_hex"]


def test_libm_mode_agrees_on_vertices():
    """oracle with the host libm (what the reference links) vs oracle with pinned_math.h (the GPU contract) on 512 stream frames
    (both variants): light blobs and the armour vertex lists -- the deliverable -- must be IDENTICAL.  `icon` / `bounding_box` go
    through the float overloads sinf / cosf / atan2f (src/core.cpp:335-337), whose last bit is the platform libm's business
    (glibc's are not correctly rounded; pinned_math.h rounds a double evaluation): there the two modes may differ, by one
    float ulp at most and in well under 1 % of the coordinates -- asserted, so that a defect in pinned_math.h (shared by the
    oracle and the kernels, i.e. common-mode for every GPU test) cannot hide behind this test."""
    from concurrent.futures import ThreadPoolExecutor

    from rmcv_amd import synth
    n = 512
    with ThreadPoolExecutor(8) as ex:
        frames = list(ex.map(lambda i: synth.frame(5000 + i, 1280, 1024, O.CAMP_BLUE, i % 2), range(n)))
    O.set_math_mode(0)
    A = [O.detect_frame(f) for f in frames]
    O.set_math_mode(1)
    B = [O.detect_frame(f) for f in frames]
    O.set_math_mode(0)
    n_arm = n_val = n_diff = 0
    for a, b in zip(A, B):
        assert a["blobs"].tobytes() == b["blobs"].tobytes()
        assert a["armours"]["vertices"].tobytes() == b["armours"]["vertices"].tobytes()
        assert np.array_equal(a["armours"]["blob_i"], b["armours"]["blob_i"]) and np.array_equal(a["armours"]["blob_j"], b["armours"]["blob_j"])
        n_arm += len(a["armours"])
        for key in ("icon", "bbox"):
            x, y = a["armours"][key].ravel(), b["armours"][key].ravel()
            n_val += x.size
            d = x != y
            n_diff += int(d.sum())
            if d.any():      # never more than one float ulp apart
                assert np.all((np.nextafter(x[d], y[d]) == y[d])), (key, x[d], y[d])
    assert n_arm > 1000
    assert n_diff * 100 < n_val, (n_diff, n_val)
