"""Independent cross-checks of the C oracle: a second, separately written restatement (pure Python / numpy / scipy)
of the integer stages must agree on random inputs.  Neither side is OpenCV (absent here); two independent readings of
the same published semantics agreeing is what backs the oracle besides the hand-derived KATs."""
import numpy as np
import pytest
from scipy import ndimage

import oracle_lib as O


def rand_binary(rng, h, w, p):
    return ((rng.random((h, w)) < p) * 255).astype(np.uint8)


@pytest.mark.parametrize("seed", range(8))
def test_morphology_vs_scipy(seed):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 60))
    a = rand_binary(rng, h, w, rng.random() * 0.6)
    se = np.ones((3, 3), bool)
    # OpenCV's default morphology border never wins: 0 for the max, 255 for the min
    d = ndimage.grey_dilation(a, footprint=se, mode="constant", cval=0)
    e = ndimage.grey_erosion(a, footprint=se, mode="constant", cval=255)
    assert np.array_equal(O.dilate3x3(a), d)
    assert np.array_equal(O.erode3x3(a), e)
    bgr = np.zeros((h, w, 3), np.uint8)
    bgr[..., 0] = a
    closed = ndimage.grey_erosion(ndimage.grey_dilation(a, footprint=se, mode="constant", cval=0), footprint=se, mode="constant", cval=255)
    assert np.array_equal(O.extract_binary(bgr, O.CAMP_BLUE, 80, O.MORPH_CLOSE), closed)


# ---- a second Suzuki-Abe (RETR_EXTERNAL, CHAIN_APPROX_NONE), written from the paper's step list on a labelled int
#      image, independent of oracle/rmcv_oracle.c (different data layout, different sweep formulation)
DIRS = [(1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1)]  # E NE N NW W SW S SE, y down


def py_find_contours(binary):
    h, w = binary.shape
    f = np.zeros((h + 2, w + 2), np.int32)
    f[1:-1, 1:-1] = (binary != 0)
    found = []
    for i in range(1, h + 1):
        lnbd = 0           # value of the last labelled pixel met on this row (0 = the frame)
        prev = 0
        j = 1
        while j <= w:
            p = int(f[i, j])
            if p != prev:
                if prev == 0 and p == 1 and not lnbd > 0:
                    # follow the outer border from (i, j); first neighbour clockwise from west
                    pts = []
                    start = (j, i)
                    s = None
                    for k in range(1, 9):
                        d = (4 - k) % 8
                        x, y = j + DIRS[d][0], i + DIRS[d][1]
                        if f[y, x] != 0:
                            s = d
                            break
                    if s is None:
                        f[i, j] = -2
                        pts.append((j - 1, i - 1))
                    else:
                        first = (j + DIRS[s][0], i + DIRS[s][1])
                        cx, cy, back = j, i, s
                        while True:
                            passed_east_zero = False
                            nd = None
                            for k in range(1, 9):
                                d = (back + k) % 8
                                x, y = cx + DIRS[d][0], cy + DIRS[d][1]
                                if f[y, x] != 0:
                                    nd = d
                                    break
                                if d == 0:
                                    passed_east_zero = True
                            if passed_east_zero:
                                f[cy, cx] = -2
                            elif f[cy, cx] == 1:
                                f[cy, cx] = 2
                            pts.append((cx - 1, cy - 1))
                            nx, ny = cx + DIRS[nd][0], cy + DIRS[nd][1]
                            if (nx, ny) == start and (cx, cy) == first:
                                break
                            cx, cy, back = nx, ny, (nd + 4) % 8
                    found.append(pts)
                    p = int(f[i, j])
                prev = p
                if p != 0 and p != 1:
                    lnbd = p
            j += 1
    return found[::-1]


@pytest.mark.parametrize("seed", range(12))
def test_contours_vs_python_suzuki(seed):
    rng = np.random.default_rng(100 + seed)
    h, w = int(rng.integers(1, 28)), int(rng.integers(1, 36))
    a = rand_binary(rng, h, w, [0.1, 0.3, 0.5, 0.7, 0.9][seed % 5])
    assert O.contours_as_lists(*O.find_contours(a)) == py_find_contours(a)


def test_contours_structured_vs_python_suzuki():
    a = np.zeros((40, 60), np.uint8)
    a[2:30, 2:40] = 255
    a[6:26, 6:36] = 0
    a[10:22, 10:32] = 255
    a[13:19, 14:28] = 0
    a[15:17, 18:22] = 255
    a[33, 1:59] = 255
    a[1:39, 50] = 255
    a[0, 0] = a[39, 59] = 255
    assert O.contours_as_lists(*O.find_contours(a)) == py_find_contours(a)


@pytest.mark.parametrize("seed", range(6))
def test_component_count_and_area(seed):
    """without nesting every 8-connected component yields exactly one external contour, whose shoelace area matches
    an independent numpy evaluation"""
    rng = np.random.default_rng(200 + seed)
    a = np.zeros((60, 80), np.uint8)
    for _ in range(12):                      # non-overlapping solid boxes: no holes, no nesting
        y, x = int(rng.integers(0, 50)), int(rng.integers(0, 70))
        a[y:y + int(rng.integers(1, 9)), x:x + int(rng.integers(1, 9))] = 255
    pts, offs = O.find_contours(a)
    _, n_comp = ndimage.label(a, structure=np.ones((3, 3)))
    assert len(offs) - 1 == n_comp
    for i in range(len(offs) - 1):
        p = pts[offs[i]:offs[i + 1]]
        x, y = p["x"].astype(np.float64), p["y"].astype(np.float64)
        shoelace = abs(np.sum(np.roll(x, 1) * y - np.roll(y, 1) * x)) / 2
        assert O.contour_area(p) == shoelace


def test_ellipse_fit_vs_numpy_least_squares():
    """the direct fit against an independent algebraic conic fit (numpy SVD) on noisy ellipse samples"""
    rng = np.random.default_rng(7)
    for _ in range(10):
        cx, cy = rng.uniform(200, 800, 2)
        A, B = rng.uniform(30, 45), rng.uniform(8, 16)
        th = rng.uniform(0, np.pi)
        t = np.linspace(0, 2 * np.pi, 48, endpoint=False)
        x = np.round(cx + A * np.cos(t) * np.cos(th) - B * np.sin(t) * np.sin(th))
        y = np.round(cy + A * np.cos(t) * np.sin(th) + B * np.sin(t) * np.cos(th))
        P = np.zeros(len(t), O.POINT)
        P["x"], P["y"] = x, y
        r, path = O.fit_ellipse_direct(P)
        # reference conic: smallest right singular vector of [x^2 xy y^2 x y 1] on centred, scaled data
        xm, ym = x - x.mean(), y - y.mean()
        s = max(np.abs(xm).max(), np.abs(ym).max())
        D = np.stack([(xm / s) ** 2, (xm / s) * (ym / s), (ym / s) ** 2, xm / s, ym / s, np.ones_like(xm)], 1)
        a, b, c, d, e, f0 = np.linalg.svd(D)[2][-1]
        den = b * b - 4 * a * c
        x0 = (2 * c * d - b * e) / den * s + x.mean()
        y0 = (2 * a * e - b * d) / den * s + y.mean()
        assert abs(r["cx"] - x0) < 0.6 and abs(r["cy"] - y0) < 0.6
        assert abs(r["h"] - 2 * A) < 0.06 * 2 * A and abs(r["w"] - 2 * B) < 0.15 * 2 * B
