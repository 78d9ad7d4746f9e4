"""the reference calls abs / atan2 / sin / cos UNQUALIFIED on floats (src/objdetect.cpp:24, 79, 131-143, 153, 157;
src/core.cpp:335-337); which functions those are depends on the headers its translation units see.  The build's default is the
float overloads; `orc_set_overload_mode` / RMCV_OPT_OVERLOADS follow a reference build that resolves them otherwise.  These are the
build's own known-answer cases (the reference ships none): inputs on which the modes MUST differ, hand-derived."""
import numpy as np
import pytest

import oracle_lib as O
from test_oracle_kat import blob


@pytest.fixture(autouse=True)
def _mode():
    O.set_math_mode(0)
    yield
    O.set_overload_mode(0)


def arm(blobs, **kw):
    return O.filter_armours(np.array(blobs, O.LIGHTBLOB), O.default_params(**kw))


def test_int_abs_truncates_every_gate_of_the_pair_loop():
    a = blob(100, 100, 40)
    # :131  angle difference 12.9 > 12 is rejected by fabsf, int abs(int) sees 12 > 12: accepted (the shear gate opened wide)
    b = blob(200, 100, 40, 12.9)
    for mode, n in ((0, 0), (1, 1), (2, 0), (3, 1)):
        O.set_overload_mode(mode)
        assert len(arm([a, b], shear_max=90.0)) == n, mode
    # :153  |dy| = 40.5 against (40 + 40) / 2 = 40: rejected as a float, 40 > 40 false once truncated
    c = blob(200, 140.5, 40)
    for mode, n in ((0, 0), (1, 1)):
        O.set_overload_mode(mode)
        assert len(arm([a, c], shear_max=90.0)) == n, mode
    # :157  |dx| = 160.75 against (40 + 40) * 2 = 160
    d = blob(260.75, 100, 40)
    for mode, n in ((0, 0), (1, 1)):
        O.set_overload_mode(mode)
        assert len(arm([a, d])) == n, mode
    # :144  shear: centre line tilted atan2(41, 100) = 22.29 deg against 22 -- |90 - 22.29| - 90 = -22.29 -> 22.29 > 22 rejected;
    # truncated: y = 41, x = 100, |(int)(90 - 22.29)| - 90 = 67 - 90 = -23 -> 23 > 22 still rejected; with 21.9 deg (40.2 / 100) the
    # float path accepts (21.9 <= 22) and the int path sees atan2(40, 100) = 21.8 -> |68| - 90 = -22 -> 22 > 22 false: accepted too
    e = blob(200, 141, 40)
    for mode in (0, 1):
        O.set_overload_mode(mode)
        assert len(arm([a, e], length_ratio_max=0.01)) == 0, mode


def test_int_abs_in_the_tilt_gate():
    """objdetect.cpp:79 -- abs(angle - 90) > tilt_max with angle - 90 = 70.6 and tilt_max = 70"""
    n = 64
    t = np.arange(n) * 2 * np.pi / n
    rot = np.radians(19.4)                                         # an ellipse lying 70.6 degrees off the vertical
    x, y = 60 * np.cos(t), 12 * np.sin(t)
    pts = np.zeros(n, O.POINT)
    pts["x"] = np.round(400 + x * np.cos(rot) - y * np.sin(rot))
    pts["y"] = np.round(300 + x * np.sin(rot) + y * np.cos(rot))
    offs = np.array([0, n], np.int32)
    ell, _ = O.fit_ellipse_direct(pts)
    tilt = abs((ell["angle"] - 90 if ell["angle"] > 90 else ell["angle"] + 90) - 90)
    assert 70 < tilt < 71, tilt                                     # the case is what its comment says
    for mode, positive in ((0, 0), (1, 1), (2, 0), (3, 1)):
        O.set_overload_mode(mode)
        blobs, _src, neg = O.filter_lightblobs(pts, offs, O.default_params(tilt_max=70.0))[:3]
        assert (len(blobs), len(neg)) == ((1, 0) if positive else (0, 1)), mode


def test_double_trig_moves_only_what_goes_through_it():
    """bit 1: `vertices` (the deliverable) do not pass through atan2 / sin / cos and stay put; `icon` does (core.cpp:335-337) and may
    move by an ulp; the pair gates see rect_angle rounded once instead of three times"""
    from rmcv_amd import synth
    diff_icon = total = 0
    for i in range(48):
        fr = synth.frame(7000 + i, 1280, 1024)
        O.set_overload_mode(0)
        a0 = O.detect_frame(fr)["armours"]
        O.set_overload_mode(2)
        a2 = O.detect_frame(fr)["armours"]
        assert len(a0) == len(a2)
        assert a0["vertices"].tobytes() == a2["vertices"].tobytes()
        total += a0["icon"].size
        diff_icon += int(np.count_nonzero(a0["icon"] != a2["icon"]))
        assert np.allclose(a0["icon"], a2["icon"], rtol=0, atol=1e-3)
    assert total > 500 and diff_icon < total // 4
