"""RMCV_OPT_OVERLOADS: the GPU follows the oracle bit for bit in every resolution of the reference's unqualified abs /
atan2 / sin / cos -- on the known-answer cases where the modes differ (tests/test_oracle_overloads.py) and on a synthetic batch."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from rmcv_amd import CAMP_BLUE, OPT_OVERLOADS, STAGE_ALL, Context, RmcvError, default_params, synth
from test_oracle_kat import blob

pytestmark = pytest.mark.gpu


@pytest.fixture()
def modes(oracle):
    yield (0, 1, 2, 3)
    oracle.set_overload_mode(0)


def test_pair_and_tilt_kats_follow_the_mode(oracle, modes):
    c = Context(device=0, max_frames=1, max_width=1280, max_height=1024)
    a = blob(100, 100, 40)
    sets = [[a, blob(200, 100, 40, 12.9)], [a, blob(200, 140.5, 40)], [a, blob(260.75, 100, 40)], [a, blob(200, 141, 40)],
            [blob(100 + 37 * k, 100 + (k % 3) * 5.5, 30 + k, 0.7 * k) for k in range(40)]]
    n = 64
    t = np.arange(n) * 2 * np.pi / n
    rot = np.radians(19.4)
    x, y = 60 * np.cos(t), 12 * np.sin(t)
    pts = np.zeros(n, oracle.POINT)
    pts["x"] = np.round(400 + x * np.cos(rot) - y * np.sin(rot))
    pts["y"] = np.round(300 + x * np.sin(rot) + y * np.cos(rot))
    offs = np.array([0, n], np.int32)
    counts = {}
    for m in modes:
        oracle.set_overload_mode(m)
        c.set_option(OPT_OVERLOADS, m)
        for k, s in enumerate(sets):
            arr = np.array(s, oracle.LIGHTBLOB)
            for kw in (dict(shear_max=90.0), {}, dict(angle_diff_max=999.0, shear_max=999.0, length_ratio_max=0.01)):
                p = oracle.default_params(**kw)
                ref = oracle.filter_armours(arr, p)
                got = c.filter_armours(arr, p.angle_diff_max, p.shear_max, p.length_ratio_max)
                assert got.tobytes() == ref.tobytes(), (m, k, kw)
                counts[(m, k, tuple(kw))] = len(ref)
        rb, _rs, rn = oracle.filter_lightblobs(pts, offs, oracle.default_params(tilt_max=70.0))[:3]
        gb, _gs, gn = c.filter_lightblobs(pts, offs, tilt_max=70.0)
        assert gb.tobytes() == rb.tobytes() and gn.tolist() == rn.tolist() and len(gb) == (m & 1), m
        lp = (1.5, 80.0, 70.0, 10.0, 99999.0, True)
        assert c.match_lightblob(pts, *lp)[0] == oracle.match_lightblob(pts, *lp)[0] == bool(m & 1), m
    assert counts[(0, 0, ("shear_max",))] == 0 and counts[(1, 0, ("shear_max",))] == 1      # the modes really differ on the KATs
    with pytest.raises(RmcvError):
        c.set_option(OPT_OVERLOADS, 4)
    c.close()


def test_full_path_batch_in_every_mode(oracle, modes):
    n, w, h = 48, 1280, 1024
    fr = synth.batch(91000, n, w, h, CAMP_BLUE, 1, threads=16)
    c = Context(device=0, max_frames=n, max_width=w, max_height=h)
    c.upload(fr)
    seen = set()
    for m in modes:
        oracle.set_overload_mode(m)
        c.set_option(OPT_OVERLOADS, m)
        c.run(default_params(), STAGE_ALL)
        c.sync()
        with ThreadPoolExecutor(16) as ex:
            refs = list(ex.map(oracle.detect_frame, fr))
        arm, offs = c.armours()
        for f in range(n):
            assert c.blobs(f)[0].tobytes() == refs[f]["blobs"].tobytes(), (m, f)
            assert arm[offs[f]:offs[f + 1]].tobytes() == refs[f]["armours"].tobytes(), (m, f)
        seen.add(arm.tobytes())
        # the per-frame chain too (its own kernels for the stage-wise calls)
        c1 = Context(device=0, max_frames=1, max_width=w, max_height=h)
        c1.set_option(OPT_OVERLOADS, m)
        for f in (0, 7):
            for _rep in range(2):                                       # the second pass runs ahead with the first one's parameters
                pts, co, _b = c1.extract_color_csr(fr[f])
                blobs, _s, _n = c1.filter_lightblobs(pts, co)
                assert blobs.tobytes() == refs[f]["blobs"].tobytes(), (m, f)
                assert c1.filter_armours(blobs).tobytes() == refs[f]["armours"].tobytes(), (m, f)
        c1.close()
    assert len(seen) >= 2                                               # (the double functions move `icon` on this batch)
    c.close()
