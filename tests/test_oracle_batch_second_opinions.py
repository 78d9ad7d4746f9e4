"""the brute-force checks of the [OCV] restatements that have no second implementation -- minAreaRect (Sklansky hull +
float rotating calipers) and solve_PnP (undistort + IPPE) -- asserted on EVERY contour and EVERY armour of a full synthetic batch
(256 frames of each stream) instead of a handful of hand-made cases.  Not a pin against OpenCV (there is none here): a property no
wrong implementation passes by accident -- the box is the minimum over all hull edges and encloses the contour; the pose
reprojects onto the image points it was computed from."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import oracle_lib as O
from rmcv_amd import synth
from test_oracle_legacy import brute_min_area
from test_oracle_pnp import project, rodrigues


def frames_of(variant, n=256):
    O.set_math_mode(0)
    with ThreadPoolExecutor(8) as ex:
        return list(ex.map(lambda i: O.detect_frame(synth.frame(60000 + 1000 * variant + i, 1280, 1024, 1, variant)), range(n)))


def test_min_area_rect_on_every_contour_of_the_batch():
    checked = 0
    for variant in (0, 1):
        for r in frames_of(variant):
            pts, offs = r["pts"], r["offs"]
            for k in range(len(offs) - 1):
                c = pts[offs[k]:offs[k + 1]]
                if len(c) < 6:
                    continue
                h = O.convex_hull(c)
                if len(h) < 3:
                    continue
                box = O.min_area_rect(c)
                w, hh, ang = float(box["w"]), float(box["h"]), np.deg2rad(float(box["angle"]))
                assert abs(w * hh - brute_min_area(c)) <= 1e-3 * max(1.0, w * hh), (variant, k)
                e = np.array([np.cos(ang), np.sin(ang)])
                nrm = np.array([-e[1], e[0]])
                d = np.stack([c["x"], c["y"]], 1).astype(float) - np.array([float(box["cx"]), float(box["cy"])])
                assert np.all(np.abs(d @ e) <= w / 2 + 1e-2) and np.all(np.abs(d @ nrm) <= hh / 2 + 1e-2), (variant, k)
                checked += 1
    assert checked > 2000


def test_solve_pnp_reprojects_every_armour_of_the_batch():
    """the detected vertices are the corners of an axis-aligned square in the image (src/core.cpp:389-399), i.e. generally NOT the
    exact image of a planar square: IPPE returns the pose of least reprojection error.  What must hold for every armour: the
    returned pose reprojects each corner to within a fraction of the square's side, the better of the two IPPE solutions was
    taken (the depth is positive and the residual is no worse than for the mirrored rotation), and the solver is deterministic."""
    cfg = O.default_pnp_config()
    n = worst = 0
    for variant in (0, 1):
        for r in frames_of(variant):
            for a in r["armours"]:
                v = a["vertices"]
                rc, rv, tv = O.solve_pnp(v, cfg)
                assert rc == 0 and tv[2] > 0
                back = project(rodrigues(rv), tv, cfg)
                side = float(np.abs(v[2] - v[0]).max())
                err = float(np.abs(back - v).max())
                assert err <= 0.02 * side + 0.05, (err, side)
                # the mirrored candidate (rotation flipped about the viewing axis of the square's centre) is not better
                rv2 = rv.copy()
                rv2[:2] = -rv2[:2]
                err2 = float(np.abs(project(rodrigues(rv2), tv, cfg) - v).max())
                assert err <= err2 + 1e-6
                rc2, rv3, tv3 = O.solve_pnp(v, cfg)
                assert np.array_equal(rv, rv3) and np.array_equal(tv, tv3)
                worst = max(worst, err / max(side, 1.0))
                n += 1
    assert n > 1000
