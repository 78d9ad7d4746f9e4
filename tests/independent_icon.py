"""A SECOND, independent restatement of the [OCV] pieces of rm::affine_correction (src/imgproc.cpp:9-35) -- getAffineTransform,
warpAffine(INTER_LINEAR, BORDER_CONSTANT) and resize(INTER_LINEAR / the exact 2:1 area case) on CV_8UC3 -- written in numpy from the
published structure of OpenCV's fixed-point paths, NOT from oracle/rmcv_oracle.c:

  * warpAffine: inverse map in 1/1024-px fixed point, quantised to 1/32 px; the four taps are blended with a 32 x 32 TABLE of
    16-bit weights that is BUILT the way OpenCV builds it (1-D float taps (1 - t, t), their float outer product scaled by 2^15,
    saturate_cast<short>, then the correction that makes every entry's four weights sum to 2^15) -- the oracle and the kernel use a
    closed form for those weights instead, so an arithmetic slip in either form shows up as a difference here;
  * resize: per-column / per-row (offset, two 11-bit coefficients) arrays from float32 arithmetic, horizontal pass into 32-bit
    rows, vertical pass with the published shifts ((b0 * (r0 >> 4)) >> 16, + 2, >> 2); whole-array numpy, no per-pixel loops.

Parity with real OpenCV stays unpinned (there is none in this image); what this removes is the single-restatement risk:
tests/test_oracle_icon_second_opinion.py compares the two on > 1000 regions, bit for bit."""
import numpy as np

AB_BITS, INTER_BITS = 10, 5
TAB = 1 << INTER_BITS
COEF_SCALE = 1 << 15


def _sat_short_round(v):
    """saturate_cast<short>(float): round half to even, clamp"""
    return np.clip(np.rint(v), -32768, 32767).astype(np.int32)


def bilinear_table():
    """itab[fy, fx, ky, kx] (int32 holding shorts): the 2 x 2 weights of fractional position (fx, fy) / 32"""
    t = np.arange(TAB, dtype=np.float32) / np.float32(TAB)
    tab1 = np.stack([np.float32(1) - t, t], axis=1)                        # [32, 2] float32
    w = (tab1[:, None, :, None] * tab1[None, :, None, :]).astype(np.float32)  # [fy, fx, ky, kx]
    itab = _sat_short_round(w * np.float32(COEF_SCALE))
    s = itab.sum(axis=(2, 3))
    # entries whose weights do not sum to 2^15 are corrected on one weight: a deficit is added to the largest weight examined,
    # a surplus taken from the smallest (the search window of the published routine collapses to the last tap for a 2 x 2 kernel)
    for fy in range(TAB):
        for fx in range(TAB):
            d = int(s[fy, fx]) - COEF_SCALE
            if d:
                itab[fy, fx, 1, 1] -= d
    assert (itab.sum(axis=(2, 3)) == COEF_SCALE).all() and itab.max() <= 32767
    return itab


_ITAB = bilinear_table()


def lu_solve(a, b):
    """cv::solve(DECOMP_LU) on a small dense system in double: partial pivoting, elimination with alpha = a[j][i] * (-1 / pivot)"""
    a = np.array(a, np.float64)
    b = np.array(b, np.float64)
    m = len(b)
    for i in range(m):
        k = i + int(np.argmax(np.abs(a[i:, i])))             # first maximum, as a linear search finds it
        if abs(a[k, i]) < np.finfo(np.float64).eps * 100:
            return None
        if k != i:
            a[[i, k], i:] = a[[k, i], i:]
            b[[i, k]] = b[[k, i]]
        d = -1.0 / a[i, i]
        for j in range(i + 1, m):
            alpha = a[j, i] * d
            a[j, i + 1:] = a[j, i + 1:] + alpha * a[i, i + 1:]
            b[j] = b[j] + alpha * b[i]
    for i in range(m - 1, -1, -1):
        s = b[i]
        for k in range(i + 1, m):
            s = s - a[i, k] * b[k]
        b[i] = s / a[i, i]
    return b


def get_affine_transform(src, dst):
    a = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        a[2 * i, 0:3] = (src[i][0], src[i][1], 1.0)
        a[2 * i + 1, 3:6] = (src[i][0], src[i][1], 1.0)
        b[2 * i], b[2 * i + 1] = dst[i]
    x = lu_solve(a, b)
    return np.zeros(6) if x is None else x


def warp_affine_same_size(roi, m):
    """cv::warpAffine(roi, dst, M, roi.size(), INTER_LINEAR, BORDER_CONSTANT, 0), roi uint8 [h, w, 3]"""
    h, w, _ = roi.shape
    m = np.array(m, np.float64)
    det = m[0] * m[4] - m[1] * m[3]
    det = 1.0 / det if det != 0 else 0.0
    a11, a22 = m[4] * det, m[0] * det
    m0, m1, m3, m4 = a11, m[1] * -det, m[3] * -det, a22
    b1 = -m0 * m[2] - m1 * m[5]
    b2 = -m3 * m[2] - m4 * m[5]
    scale = float(1 << AB_BITS)
    rd = (1 << AB_BITS) // TAB // 2
    ys = np.arange(h, dtype=np.float64)
    xs = np.arange(w, dtype=np.float64)
    X0 = np.rint((m1 * ys + b1) * scale).astype(np.int64) + rd
    Y0 = np.rint((m4 * ys + b2) * scale).astype(np.int64) + rd
    ad = np.rint(m0 * xs * scale).astype(np.int64)
    bd = np.rint(m3 * xs * scale).astype(np.int64)
    X = (X0[:, None] + ad[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bd[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    wts = _ITAB[(Y & (TAB - 1)), (X & (TAB - 1))].astype(np.int64)       # [h, w, 2, 2]
    pad = np.zeros((h + 2, w + 2, 3), np.int64)                            # constant border 0: one ring is enough after clipping
    pad[1:-1, 1:-1] = roi
    acc = np.zeros((h, w, 3), np.int64)
    for ky in range(2):
        for kx in range(2):
            yy = np.clip(sy + ky, -1, h) + 1
            xx = np.clip(sx + kx, -1, w) + 1
            acc += pad[yy, xx] * wts[:, :, ky, kx][..., None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def resize_to(src, dw, dh):
    """cv::resize(src, dst, (dw, dh)) with the default INTER_LINEAR on CV_8UC3"""
    sh, sw, _ = src.shape
    s = src.astype(np.int64)
    if sw == 2 * dw and sh == 2 * dh:                                      # the exact 2:1 case is computed as a 2 x 2 box average
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def taps(dn, sn, clamp_both):
        scale = float(sn) / dn
        f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        o = np.floor(f).astype(np.int64)
        f = (f - o.astype(np.float32)).astype(np.float32)
        if clamp_both:
            lo = o < 0
            f[lo], o[lo] = 0, 0
            hi = o >= sn - 1
            f[hi], o[hi] = 0, sn - 1
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)
        c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return o, c0, c1
    xo, xa0, xa1 = taps(dw, sw, True)
    yo, yb0, yb1 = taps(dh, sh, False)
    x1 = np.minimum(xo + 1, sw - 1)                                         # (a clamped column has weight 0 on its second tap)
    rows = s[:, xo] * xa0[None, :, None] + s[:, x1] * xa1[None, :, None]    # horizontal pass: [sh, dw, 3], 11-bit scaled
    y0 = np.clip(yo, 0, sh - 1)
    y1 = np.clip(yo + 1, 0, sh - 1)
    v = (((yb0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((yb1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def affine_correction(img, icon, side=20):
    """rm::affine_correction(img, icon, {side, side}) -> (out uint8 [side, side, 3] or zeros, icon clamped in place (float32 [4, 2]), rc)"""
    h, w, _ = img.shape
    icon = np.array(icon, np.float32)
    icon[:, 0] = np.maximum(np.float32(0), np.minimum(icon[:, 0], np.float32(w) - np.float32(1)))
    icon[:, 1] = np.maximum(np.float32(0), np.minimum(icon[:, 1], np.float32(h) - np.float32(1)))
    pi = np.rint(icon.astype(np.float64)).astype(np.int64)                  # Point2f -> Point: cvRound
    bx, by = int(pi[:, 0].min()), int(pi[:, 1].min())
    bw, bh = int(pi[:, 0].max()) - bx + 1, int(pi[:, 1].max()) - by + 1
    out = np.zeros((side, side, 3), np.uint8)
    if bw <= 0 or bh <= 0 or bx < 0 or by < 0 or bx + bw > w or by + bh > h:
        return out, icon, 1
    f32 = np.float32
    src = [(icon[1, 0] - f32(bx), icon[1, 1] - f32(by)), (icon[2, 0] - f32(bx), icon[2, 1] - f32(by)), (icon[0, 0] - f32(bx), icon[0, 1] - f32(by))]
    dst = [(0.0, 0.0), (float(bw), 0.0), (0.0, float(bh))]
    m = get_affine_transform([(float(a), float(b)) for a, b in src], dst)
    warped = warp_affine_same_size(img[by:by + bh, bx:bx + bw], m)
    return resize_to(warped, side, side), icon, 0
