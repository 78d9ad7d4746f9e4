"""The oracle's ONE deliberate deviation from OpenCV's structure, checked against an independent solver.

cv::fitEllipseDirect falls back to the general ("LIN") conic fit when its own result is wild or its 3x3 system is singular
(oracle/rmcv_oracle.c: fit_ellipse_general).  OpenCV solves the fallback's two least-squares systems (n x 5, n x 3) and the
2 x 2 centre system with a Jacobi SVD of the design matrix (`cv::solve(..., DECOMP_SVD)`); the oracle -- and the device twin,
rmcv_amd/csrc/device_fit.h -- solve the normal equations with a cyclic Jacobi and the centre by Cramer's rule, so that a
contour is one pass over its points.  That squares the condition number, so "agrees to double rounding" is a claim that needs
evidence: here every general-fit contour of 512 stream frames (both variants, ~500 contours) is re-solved with LAPACK's
SVD least squares (numpy.linalg.lstsq) on the full design matrices and the resulting float32 RotatedRect is compared bit for
bit, together with the gate decisions rm::filter_lightblobs takes on it (src/objdetect.cpp:70-80).

This branch is not rare on the synthetic stream: roughly one fitted contour in six takes it (bars longer than ~200 points).
"""
import math
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import oracle_lib as O

FLT_EPSILON = float(np.finfo(np.float32).eps)
f32 = np.float32


def general_fit_svd(pts):
    """fitEllipseNoDirect restated with SVD least squares on the n x k design matrices (the structure OpenCV uses)"""
    n = len(pts)
    x, y = pts["x"].astype(f32), pts["y"].astype(f32)
    cx = f32(x.sum(dtype=f32) / f32(n))              # integer sums below 2^24: exact in any order
    cy = f32(y.sum(dtype=f32) / f32(n))
    fx, fy = (x - cx).astype(f32), (y - cy).astype(f32)
    s = float(np.cumsum(np.abs(fx.astype(np.float64)) + np.abs(fy.astype(np.float64)))[-1])   # sequential, like the oracle
    scale = 100.0 / max(s, FLT_EPSILON)
    eps = f32(0)

    def design(eps):
        if eps != 0:
            i = np.arange(n)
            ox = ((i & 1) * 2 - 1).astype(f32) * eps
            oy = ((i & 2) - 1).astype(f32) * eps
            gx, gy = ((x + ox).astype(f32) - cx).astype(f32), ((y + oy).astype(f32) - cy).astype(f32)
        else:
            gx, gy = fx, fy
        return gx.astype(np.float64) * scale, gy.astype(np.float64) * scale

    for it in range(2):
        px, py = design(eps)
        A = np.stack([-px * px, -py * py, -px * py, px, py], 1)
        gfp, _, _, sv = np.linalg.lstsq(A, np.full(n, 10000.0), rcond=None)
        if it == 0 and sv.max() * FLT_EPSILON > sv.min():
            eps = f32(s / (n * 2) * 1e-3)
            continue
        break
    A2 = np.array([[2 * gfp[0], gfp[2]], [gfp[2], 2 * gfp[1]]])
    rp = np.zeros(5)
    if np.linalg.det(A2) != 0.0:
        rp[:2] = np.linalg.lstsq(A2, np.array([gfp[3], gfp[4]]), rcond=None)[0]
    px, py = design(eps)
    A3 = np.stack([(px - rp[0]) ** 2, (py - rp[1]) ** 2, (px - rp[0]) * (py - rp[1])], 1)
    g3 = np.linalg.lstsq(A3, np.ones(n), rcond=None)[0]
    min_eps = 1e-8
    rp[4] = -0.5 * math.atan2(g3[2], g3[1] - g3[0])
    t = g3[2] / math.sin(-2.0 * rp[4]) if abs(g3[2]) > min_eps else g3[1] - g3[0]
    rp[2] = abs(g3[0] + g3[1] - t)
    if rp[2] > min_eps:
        rp[2] = math.sqrt(2.0 / rp[2])
    rp[3] = abs(g3[0] + g3[1] + t)
    if rp[3] > min_eps:
        rp[3] = math.sqrt(2.0 / rp[3])
    out = np.zeros(1, O.RRECT)[0]
    out["cx"] = f32(f32(rp[0] / scale) + cx)
    out["cy"] = f32(f32(rp[1] / scale) + cy)
    w, h, ang = f32(rp[2] * 2 / scale), f32(rp[3] * 2 / scale), f32(0)
    if w > h:
        w, h = h, w
        ang = f32(90 + rp[4] * 180 / math.pi)
    if ang < -180:
        ang = f32(ang + 360)
    if ang > 360:
        ang = f32(ang - 360)
    out["w"], out["h"], out["angle"] = w, h, ang
    return out


def gates(e, tilt_max=70.0, ratio=(1.5, 80.0)):
    """the decisions of src/objdetect.cpp:70-80 on a fitted ellipse: (ratio test, tilt test)"""
    r = f32(e["h"]) / f32(e["w"]) if e["w"] > 0 else f32(np.inf)
    ang = f32(e["angle"] - 90) if e["angle"] > 90 else f32(e["angle"] + 90)      # lightblob ctor, src/core.cpp:10-13
    return bool(ratio[0] <= r <= ratio[1]), bool(abs(f32(ang - 90)) <= tilt_max)


def general_fit_contours(first, n_frames, variant):
    from rmcv_amd import synth

    def one(i):
        f = synth.frame(first + i, 1280, 1024, O.CAMP_BLUE, variant)
        r = O.detect_frame(f)
        out = []
        for k in range(len(r["offs"]) - 1):
            p = r["pts"][r["offs"][k]:r["offs"][k + 1]]
            if len(p) >= 6:
                e, path = O.fit_ellipse_direct(p)
                out.append((p.copy(), e.copy(), path))
        return out
    with ThreadPoolExecutor(8) as ex:
        return [c for frame in ex.map(one, range(n_frames)) for c in frame]


def test_general_fit_equals_an_svd_least_squares_solve():
    O.set_math_mode(0)
    fitted = general_count = differ = gate_differ = 0
    worst = None
    for variant in (0, 1):
        for p, e, path in general_fit_contours(0 if variant == 0 else 200000, 256, variant):
            fitted += 1
            if path != 1:
                continue
            general_count += 1
            ref = general_fit_svd(p)
            if ref.tobytes() != e.tobytes():
                differ += 1
                worst = worst or (len(p), e, ref)
            if gates(ref) != gates(e):
                gate_differ += 1
    assert general_count >= 200 and general_count * 20 > fitted, (general_count, fitted)   # the branch is common on this stream
    assert gate_differ == 0
    assert differ == 0, "%d of %d general-fit contours differ from the SVD solve in some float32 bit, e.g. %s" % (differ, general_count, worst)


def test_general_fit_on_thin_bars_admitted_by_the_ratio_gate():
    """is_good_box fails beyond aspect 30 while rm::filter_lightblobs admits ratios up to 80 (executable/main.cpp:174): thin
    bars are exactly the contours whose ellipse comes from the fallback.  Synthetic 2..4 px wide bars at many tilts."""
    O.set_math_mode(0)
    rng = np.random.default_rng(11)
    checked = 0
    for _ in range(60):
        img = np.zeros((420, 420), np.uint8)
        L, wd, th = int(rng.integers(90, 190)), int(rng.integers(2, 5)), math.radians(float(rng.uniform(-35, 35)))
        yy, xx = np.mgrid[0:420, 0:420]
        u = (xx - 210) * math.cos(th) + (yy - 210) * math.sin(th)
        v = -(xx - 210) * math.sin(th) + (yy - 210) * math.cos(th)
        img[(np.abs(v) <= L) & (np.abs(u) <= wd / 2)] = 255
        pts, offs = O.find_contours(img)
        for k in range(len(offs) - 1):
            p = pts[offs[k]:offs[k + 1]]
            if len(p) < 6:
                continue
            e, path = O.fit_ellipse_direct(p)
            if path == 1:
                ref = general_fit_svd(p)
                assert ref.tobytes() == e.tobytes(), (L, wd, math.degrees(th), e, ref)
                assert gates(ref) == gates(e)
                checked += 1
    assert checked >= 30
