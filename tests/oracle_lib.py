"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Nothing under rmcv_amd/ may import this module; it is
used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "liboracle.so")

POINT = np.dtype([("x", "<i4"), ("y", "<i4")])
RRECT = np.dtype([("cx", "<f4"), ("cy", "<f4"), ("w", "<f4"), ("h", "<f4"), ("angle", "<f4")])
LIGHTBLOB = np.dtype([("angle", "<f4"), ("target", "<i4"), ("center", "<f4", (2,)),
                      ("vertices", "<f4", (4, 2)), ("size", "<f4", (2,))])
ARMOUR = np.dtype([("icon", "<f4", (4, 2)), ("vertices", "<f4", (4, 2)), ("bbox", "<f4", (4,)),
                   ("blob_i", "<i4"), ("blob_j", "<i4")])
assert LIGHTBLOB.itemsize == 56 and ARMOUR.itemsize == 88

CAMP_RED, CAMP_BLUE, CAMP_GUIDELIGHT, CAMP_NEUTRAL = 0, 1, 2, -1
MORPH_NONE, MORPH_DILATE, MORPH_CLOSE = 0, 1, 2


class Params(C.Structure):
    _fields_ = [("camp", C.c_int32), ("lower_bound", C.c_int32), ("morph", C.c_int32), ("tilt_max", C.c_float),
                ("ratio_lo", C.c_float), ("ratio_hi", C.c_float), ("area_lo", C.c_double), ("area_hi", C.c_double),
                ("angle_diff_max", C.c_float), ("shear_max", C.c_float), ("length_ratio_max", C.c_float),
                ("_pad", C.c_int32)]


def build():
    subprocess.run(["make", "-s", "-C", os.path.join(_ROOT, "oracle")], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_contour_area.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_math_mode(mode):
    lib().orc_set_math_mode(int(mode))


def set_overload_mode(mode):
    """bit 0 = the reference's unqualified abs(float) is int abs(int); bit 1 = its atan2 / sin / cos are the double functions"""
    lib().orc_set_overload_mode(int(mode))


def default_params(**kw):
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def extract_binary(bgr, camp=CAMP_BLUE, lower_bound=80, morph=MORPH_CLOSE):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    rc = lib().orc_extract_binary(_p(bgr), w, h, 3 * w, camp, lower_bound, morph, _p(out))
    assert rc == 0, rc
    return out


def dilate3x3(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().orc_dilate3x3(_p(img), _p(out), img.shape[1], img.shape[0])
    return out


def erode3x3(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().orc_erode3x3(_p(img), _p(out), img.shape[1], img.shape[0])
    return out


def find_contours(binary, cap_pts=None, cap_contours=None):
    """returns (points[n_points] POINT, offs[n_contours+1] int32)"""
    binary = np.ascontiguousarray(binary, np.uint8)
    h, w = binary.shape
    cap_pts = cap_pts or max(16, 4 * int(np.count_nonzero(binary)) + 16)
    cap_contours = cap_contours or max(16, int(np.count_nonzero(binary)) + 16)
    pts = np.zeros(cap_pts, POINT)
    offs = np.zeros(cap_contours + 1, np.int32)
    nc, npnt = C.c_int32(0), C.c_int32(0)
    rc = lib().orc_find_contours(_p(binary), w, h, _p(pts), cap_pts, _p(offs), cap_contours, C.byref(nc), C.byref(npnt))
    assert rc == 0, rc
    return pts[:npnt.value].copy(), offs[:nc.value + 1].copy()


def contours_as_lists(pts, offs):
    return [[(int(p["x"]), int(p["y"])) for p in pts[offs[i]:offs[i + 1]]] for i in range(len(offs) - 1)]


def contour_area(pts):
    pts = np.ascontiguousarray(pts, POINT)
    return lib().orc_contour_area(_p(pts), len(pts))


def fit_ellipse_direct(pts):
    pts = np.ascontiguousarray(pts, POINT)
    out = np.zeros(1, RRECT)
    path = lib().orc_fit_ellipse_direct(_p(pts), len(pts), _p(out))
    return out[0], path


def make_lightblob(rrect, camp):
    r = np.zeros(1, RRECT)
    r[0] = rrect
    out = np.zeros(1, LIGHTBLOB)
    lib().orc_make_lightblob(_p(r), camp, _p(out))
    return out[0]


def make_armour(a, b):
    aa = np.zeros(1, LIGHTBLOB)
    bb = np.zeros(1, LIGHTBLOB)
    aa[0], bb[0] = a, b
    out = np.zeros(1, ARMOUR)
    lib().orc_make_armour(_p(aa), _p(bb), _p(out))
    return out[0]


def filter_lightblobs(pts, offs, p=None, want_ellipses=False):
    p = p or default_params()
    pts = np.ascontiguousarray(pts, POINT)
    offs = np.ascontiguousarray(offs, np.int32)
    n = len(offs) - 1
    blobs = np.zeros(max(n, 1), LIGHTBLOB)
    src = np.zeros(max(n, 1), np.int32)
    neg = np.zeros(max(n, 1), np.int32)
    ell = np.zeros(max(n, 1), RRECT)
    nb, nn = C.c_int32(0), C.c_int32(0)
    rc = lib().orc_filter_lightblobs(_p(pts), _p(offs), n, C.c_float(p.tilt_max), C.c_float(p.ratio_lo),
                                     C.c_float(p.ratio_hi), C.c_double(p.area_lo), C.c_double(p.area_hi), p.camp,
                                     _p(blobs), len(blobs), C.byref(nb), _p(src), _p(neg), C.byref(nn), _p(ell))
    assert rc == 0, rc
    if want_ellipses:
        return blobs[:nb.value].copy(), src[:nb.value].copy(), neg[:nn.value].copy(), ell[:nb.value].copy()
    return blobs[:nb.value].copy(), src[:nb.value].copy(), neg[:nn.value].copy()


def filter_armours(blobs, p=None):
    p = p or default_params()
    blobs = np.ascontiguousarray(blobs, LIGHTBLOB)
    n = len(blobs)
    cap = max(1, n * (n - 1) // 2)
    out = np.zeros(cap, ARMOUR)
    na = C.c_int32(0)
    rc = lib().orc_filter_armours(_p(blobs), n, C.c_float(p.angle_diff_max), C.c_float(p.shear_max),
                                  C.c_float(p.length_ratio_max), p.camp, _p(out), cap, C.byref(na))
    assert rc == 0, rc
    return out[:na.value].copy()


def detect_frame(bgr, p=None, cap_pts=1 << 18, cap_contours=1 << 14, cap_blobs=4096, cap_armours=4096):
    """whole per-frame path; returns dict(binary, pts, offs, blobs, armours)"""
    p = p or default_params()
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    binary = np.empty((h, w), np.uint8)
    pts = np.zeros(cap_pts, POINT)
    offs = np.zeros(cap_contours + 1, np.int32)
    blobs = np.zeros(cap_blobs, LIGHTBLOB)
    arm = np.zeros(cap_armours, ARMOUR)
    nc, nb, na = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    rc = lib().orc_detect_frame(_p(bgr), w, h, 3 * w, C.byref(p), _p(binary), _p(pts), cap_pts, _p(offs), cap_contours,
                                C.byref(nc), _p(blobs), cap_blobs, C.byref(nb), _p(arm), cap_armours, C.byref(na))
    assert rc == 0, rc
    o = offs[:nc.value + 1].copy()
    return dict(binary=binary, pts=pts[:o[-1] if nc.value else 0].copy(), offs=o, blobs=blobs[:nb.value].copy(),
                armours=arm[:na.value].copy())


def affine_correction(bgr, icon):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    ic = np.ascontiguousarray(icon, np.float32).copy()
    out = np.zeros((20, 20, 3), np.uint8)
    rc = lib().orc_affine_correction(_p(bgr), w, h, 3 * w, _p(ic), _p(out))
    return out, ic, rc


def classify_armours(bgr, armours, svm):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    arm = np.ascontiguousarray(armours, ARMOUR).copy()
    n = len(arm)
    wts, rho, labels = svm
    ident = np.zeros(max(n, 1), np.int32)
    icons = np.zeros((max(n, 1), 20, 20, 3), np.uint8)
    lib().orc_classify_armours(_p(bgr), w, h, 3 * w, _p(arm), n, _p(np.ascontiguousarray(wts, np.float32)),
                               _p(np.ascontiguousarray(rho, np.float64)), _p(np.ascontiguousarray(labels, np.int32)),
                               len(labels), _p(ident), _p(icons))
    return ident[:n].copy(), arm, icons[:n].copy()


# ---------------------------------------------------------------- legacy matcher (oracle/rmcv_oracle_legacy.c)
def convex_hull(pts, pruned=False):
    pts = np.ascontiguousarray(pts, POINT)
    out = np.zeros(max(4 * len(pts), 4), np.int32)
    fn = lib().orc_convex_hull_pruned if pruned else lib().orc_convex_hull
    n = fn(_p(pts), len(pts), _p(out))
    return None if n < 0 else out[:n].copy()


def min_area_rect(pts):
    pts = np.ascontiguousarray(pts, POINT)
    out = np.zeros(1, RRECT)
    lib().orc_min_area_rect(_p(pts), len(pts), _p(out))
    return out[0]


def match_lightblob(pts, min_ratio, max_ratio, tilt_angle, min_area, max_area, fit_ellipse):
    pts = np.ascontiguousarray(pts, POINT)
    out = np.zeros(1, RRECT)
    ok = lib().orc_match_lightblob(_p(pts), len(pts), C.c_float(min_ratio), C.c_float(max_ratio), C.c_float(tilt_angle),
                                   C.c_float(min_area), C.c_float(max_area), int(fit_ellipse), _p(out))
    return bool(ok), out[0]


def find_lightblobs(bgr, pts, offs, min_ratio, max_ratio, tilt_angle, min_area, max_area, fit_ellipse):
    """returns (blobs, blob_src, boxes)"""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    pts = np.ascontiguousarray(pts, POINT)
    offs = np.ascontiguousarray(offs, np.int32)
    n = len(offs) - 1
    blobs = np.zeros(max(n, 1), LIGHTBLOB)
    src = np.zeros(max(n, 1), np.int32)
    boxes = np.zeros(max(n, 1), RRECT)
    nb = C.c_int32(0)
    rc = lib().orc_find_lightblobs(_p(bgr), w, h, 3 * w, _p(pts), _p(offs), n, C.c_float(min_ratio), C.c_float(max_ratio),
                                   C.c_float(tilt_angle), C.c_float(min_area), C.c_float(max_area), int(fit_ellipse),
                                   _p(blobs), len(blobs), C.byref(nb), _p(src), _p(boxes))
    assert rc == 0, rc
    return blobs[:nb.value].copy(), src[:nb.value].copy(), boxes[:nb.value].copy()


def lightblob_overlap(blobs, left, right):
    blobs = np.ascontiguousarray(blobs, LIGHTBLOB)
    return lib().orc_lightblob_overlap(_p(blobs), len(blobs), int(left), int(right))


# ---------------------------------------------------------------- solve_PnP (oracle/rmcv_oracle_pnp.c)
class PnpConfig(C.Structure):
    _fields_ = [("camera_matrix", C.c_double * 9), ("dist", C.c_double * 5), ("gripper2camera", C.c_double * 16),
                ("square_w", C.c_float), ("square_h", C.c_float)]


def default_pnp_config():
    c = PnpConfig()
    lib().orc_default_pnp_config(C.byref(c))
    return c


def solve_pnp(vertices, cfg=None):
    """vertices: 4x2 float32 (armour.vertices).  returns (rc, rvec[3], tvec[3])"""
    cfg = cfg or default_pnp_config()
    v = np.ascontiguousarray(vertices, np.float32).reshape(4, 2)
    r, t = np.zeros(3), np.zeros(3)
    rc = lib().orc_solve_pnp(_p(v), C.byref(cfg), _p(r), _p(t))
    return rc, r, t


def locate_armours(armours, cfg=None, base2gripper=None):
    """returns (rvecs[n,3], tvecs[n,3], positions[n,3])"""
    cfg = cfg or default_pnp_config()
    arm = np.ascontiguousarray(armours, ARMOUR)
    n = len(arm)
    r, t, p = np.zeros((max(n, 1), 3)), np.zeros((max(n, 1), 3)), np.zeros((max(n, 1), 3))
    b = None if base2gripper is None else np.ascontiguousarray(base2gripper, np.float64).reshape(16)
    lib().orc_locate_armours(_p(arm), n, C.byref(cfg), _p(b) if b is not None else None, _p(r), _p(t), _p(p))
    return r[:n], t[:n], p[:n]


# ---------------------------------------------------------------- observable tracker parts (oracle/rmcv_oracle_track.c)
def max_iou(self_armour, armours):
    me = np.ascontiguousarray(self_armour, ARMOUR).reshape(1)
    arr = np.ascontiguousarray(armours, ARMOUR)
    idx, iou = C.c_int32(0), C.c_float(0)
    lib().orc_max_iou(_p(me), _p(arr), len(arr), C.byref(idx), C.byref(iou))
    return idx.value, iou.value


def identity_max(history):
    ids = np.array(sorted(history), np.int32)
    cnt = np.array([history[int(k)] for k in ids], np.int32)
    mid, pr = C.c_int32(0), C.c_double(0)
    lib().orc_identity_max(_p(ids), _p(cnt), len(ids), C.byref(mid), C.byref(pr))
    return mid.value, pr.value


# ---- tracker state: the oracle's twin of rmcv_track_* ----------------------------------------------------------
def _track_dtype():
    from rmcv_amd import abi
    return abi.TRACK


def track_new(armour, identity, timestamp, position, noise=(5e-5, 0.5, 0.05)):
    t = np.zeros(1, _track_dtype())
    a = np.ascontiguousarray(armour, ARMOUR).reshape(1)
    pos = np.ascontiguousarray(position, np.float64)
    lib().orc_track_init(_p(t), _p(a), int(identity), C.c_int64(int(timestamp)), _p(pos))
    if noise is not None:
        lib().orc_track_reset(_p(t), C.c_double(noise[0]), C.c_double(noise[1]), C.c_double(noise[2]))
    return t[0]


def track_update(track, observation, tick_frequency=1e9):
    t = np.array([track], _track_dtype())
    o = np.array([observation], _track_dtype())
    assert lib().orc_track_update(_p(t), _p(o), C.c_double(tick_frequency)) == 0
    return t[0]


def track_predict(track, new_timestamp, tick_frequency=1e9):
    t = np.array([track], _track_dtype())
    assert lib().orc_track_predict(_p(t), C.c_int64(int(new_timestamp)), C.c_double(tick_frequency)) == 0
    return t[0]


def track_step(tracking, observations, cap=64, tick_frequency=1e9):
    buf = np.zeros(cap, _track_dtype())
    nt = C.c_int32(len(tracking))
    if len(tracking):
        buf[:len(tracking)] = tracking
    obs = np.array(observations, _track_dtype()).copy() if len(observations) else np.zeros(1, _track_dtype())
    no = C.c_int32(len(observations))
    assert lib().orc_track_step(_p(buf), C.byref(nt), cap, _p(obs), C.byref(no), C.c_double(tick_frequency)) == 0
    return buf[:nt.value].copy()
