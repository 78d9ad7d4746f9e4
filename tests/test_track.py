"""SURVEY 8f-4, the tracker: rm::armour::max_IoU / identity_max (src/core.cpp:124-162), the filter state of reset / update
(src/core.cpp:51-122) and the association loop of the tracking thread (executable/main.cpp:57-88).  Host-side functions of the
C-ABI (no device work), checked against the oracle, against an independent numpy Kalman step and against known answers."""
import math

import numpy as np
import pytest

import oracle_lib as O
from rmcv_amd.abi import RmcvError
from rmcv_amd.api import Context


@pytest.fixture(autouse=True, params=[pytest.param("host"), pytest.param("gpu-box", marks=pytest.mark.gpu)])
def where(request):
    """every test of this module runs twice: in the CPU suite (`-m "not gpu"`; the tracker is host code of librmcv_hip.so and
    needs no device) AND in the driver's `-m gpu` run, i.e. against the library as deployed on the GPU box"""
    return request.param


def arm(x, y, w, h):
    a = np.zeros(1, O.ARMOUR)
    a[0]["bbox"] = (x, y, w, h)
    return a[0]


def test_max_iou_known_answers():
    me = arm(10, 10, 20, 20)
    lst = np.array([arm(100, 100, 5, 5), arm(20, 10, 20, 20), arm(10, 10, 20, 20), arm(12, 12, 4, 4)], O.ARMOUR)
    for f in (O.max_iou, Context.max_iou):
        idx, iou = f(me, lst)
        assert idx == 2 and iou == 1.0
        idx, iou = f(me, lst[:2])
        assert idx == 1 and abs(iou - 200.0 / 600.0) < 1e-7      # half overlap: 200 / (400 + 400 - 200)
        assert f(me, lst[:1]) == (-1, 0.0)
        assert f(me, lst[:0]) == (-1, 0.0)
        assert f(arm(0, 0, 0, 0), lst) [0] == -1                  # an empty box overlaps nothing (0/positive = 0, never > 0)


def test_max_iou_random_against_oracle():
    rng = np.random.default_rng(4)
    for _ in range(200):
        n = int(rng.integers(0, 12))
        lst = np.zeros(n, O.ARMOUR)
        lst["bbox"] = np.stack([rng.uniform(-50, 600, n), rng.uniform(-50, 500, n), rng.uniform(0, 120, n), rng.uniform(0, 120, n)], 1)
        me = arm(*rng.uniform(-50, 500, 2), *rng.uniform(1, 150, 2))
        a, b = O.max_iou(me, lst), Context.max_iou(me, lst)
        assert a[0] == b[0] and np.float32(a[1]).tobytes() == np.float32(b[1]).tobytes()


def test_identity_max():
    for f in (O.identity_max, Context.identity_max):
        assert f({}) == (-1, 0.0)
        mid, p = f({3: 5})
        assert mid == 3 and p == 1.0
        mid, p = f({1: 2, 4: 2, 7: 1})                             # tie: the lower key comes first in std::map order
        assert mid == 1 and abs(p - math.exp(2) / (2 * math.exp(2) + math.exp(1))) < 1e-15
        mid, p = f({5: 1, 2: 9, 8: 3})
        assert mid == 2
    hist = {int(k): int(v) for k, v in zip(range(0, 14, 2), [3, 1, 4, 1, 5, 9, 2])}
    assert O.identity_max(hist) == Context.identity_max(hist)


# ---------------------------------------------------------------- filter state: reset / update / predict
TICK = 1e9


def obs_at(pos, stamp, identity=3, box=(100, 100, 40, 40), f=Context.track_new):
    return f(arm(*box), identity, stamp, pos)


def test_reset_and_first_update_are_the_references():
    for new, upd in ((Context.track_new, Context.track_update), (O.track_new, O.track_update)):
        t = obs_at((1.0, 2.0, 3.0), 1000, f=new)
        assert np.array_equal(t["measurement_matrix"], np.eye(6)) and np.array_equal(t["error_cov_post"], 0.05 * np.eye(6))
        assert np.array_equal(t["process_noise_cov"], 5e-5 * np.eye(6)) and np.array_equal(t["measurement_noise_cov"], 0.5 * np.eye(6))
        A = np.eye(6)
        A[0, 3] = A[1, 4] = A[2, 5] = 1.0
        assert np.array_equal(t["transition"], A) and t["initialized"] == 0 and not t["error_cov_pre"].any()
        # first observation (src/core.cpp:98-106): a correction WITHOUT a prediction -- errorCovPre is still zero, so the gain is
        # zero, the state stays zero and errorCovPost becomes zero: mirrored, not "fixed"
        u = upd(t, obs_at((1.0, 2.0, 3.0), 2000, f=new), TICK)
        assert u["initialized"] == 1 and u["timestamp"] == 2000 and not u["state_post"].any() and not u["error_cov_post"].any()
        assert u["measurement"].tolist() == [1.0, 2.0, 3.0, 0, 0, 0] and u["n_ids"] == 1 and u["ids"][0] == 3 and u["counts"][0] == 1


def test_constant_velocity_target_recovers_position_and_velocity():
    v = np.array([0.8, -0.3, 0.05])                  # units per second
    p0 = np.array([10.0, 5.0, 2.0])
    dt_ticks = 10_000_000                            # 10 ms at 1e9 ticks/s
    for new, upd in ((Context.track_new, Context.track_update), (O.track_new, O.track_update)):
        # the reference's first update zeroes the state AND its covariance (see above), so the filter starts out trusting a zero
        # state and only the process noise (5e-5) lets it move: it needs ~1000 observations to lock on -- then it is exact
        t = obs_at(p0, 0, f=new)
        n = 2000
        for k in range(1, n + 1):
            t = upd(t, obs_at(p0 + v * (k * dt_ticks / TICK), k * dt_ticks, identity=3 if k % 7 else 5, f=new), TICK)
        pos = p0 + v * (n * dt_ticks / TICK)
        assert np.allclose(t["state_post"][:3], pos, atol=1e-6), (t["state_post"], pos)
        assert np.allclose(t["state_post"][3:], v, atol=1e-6), t["state_post"]
        assert t["n_ids"] == 2 and t["ids"][:2].tolist() == [3, 5] and t["counts"][:2].sum() == n
        # src/core.cpp:127 sums exp(count): past 709 observations of one identity that is +inf, every probability NaN, and
        # identity_max answers -1 -- the reference's behaviour, mirrored by product and oracle alike
        hist = {int(i): int(c) for i, c in zip(t["ids"][:2], t["counts"][:2])}
        assert hist[3] > 709 and Context.identity_max(hist)[0] == -1 and O.identity_max(hist)[0] == -1
        assert Context.identity_max({3: 600, 5: 100})[0] == 3


def test_update_equals_an_independent_numpy_kalman_step():
    """predict + correct with numpy's own linear algebra (np.linalg.solve), to rounding"""
    rng = np.random.default_rng(3)
    t = obs_at((0.0, 0.0, 0.0), 0)
    t = Context.track_update(t, obs_at((0.1, 0.2, 0.3), 5_000_000), TICK)
    for k in range(2, 30):
        z = rng.normal(size=3)
        o = obs_at(z, k * 7_000_000)
        n = Context.track_update(t, o, TICK)
        dt = (int(o["timestamp"]) - int(t["timestamp"])) / TICK
        A = t["transition"].copy()
        A[0, 3] = A[1, 4] = A[2, 5] = dt
        x_pre = A @ t["state_post"]
        P_pre = A @ t["error_cov_post"] @ A.T + t["process_noise_cov"]
        meas = np.concatenate([z, (z - t["measurement"][:3]) / dt])
        H, R = t["measurement_matrix"], t["measurement_noise_cov"]
        S = H @ P_pre @ H.T + R
        Kg = np.linalg.solve(S, H @ P_pre).T
        x_post = x_pre + Kg @ (meas - H @ x_pre)
        P_post = P_pre - Kg @ (H @ P_pre)
        assert np.allclose(n["state_post"], x_post, rtol=1e-10, atol=1e-12)
        assert np.allclose(n["error_cov_post"], P_post, rtol=1e-9, atol=1e-13)
        assert np.allclose(n["gain"], Kg, rtol=1e-9, atol=1e-13)
        t = n


def test_filter_state_product_equals_oracle_bit_for_bit():
    rng = np.random.default_rng(9)
    a = obs_at((1.0, 1.0, 1.0), 0, f=Context.track_new)
    b = obs_at((1.0, 1.0, 1.0), 0, f=O.track_new)
    assert a.tobytes() == b.tobytes()
    stamp = 0
    for k in range(200):
        stamp += int(rng.integers(1_000_000, 30_000_000))
        if rng.random() < 0.25:                      # a missed frame: coast
            a, b = Context.track_predict(a, stamp, TICK), O.track_predict(b, stamp, TICK)
        else:
            pos, ident = rng.normal(size=3) * 3, int(rng.integers(0, 7))
            a = Context.track_update(a, obs_at(pos, stamp, ident, f=Context.track_new), TICK)
            b = O.track_update(b, obs_at(pos, stamp, ident, f=O.track_new), TICK)
        assert a.tobytes() == b.tobytes(), k
    assert a["n_ids"] >= 5 and np.isfinite(a["state_post"]).all()


# ---------------------------------------------------------------- the tracking thread's loop (main.cpp:60-85)
def test_association_loop_known_scenarios():
    for step, new in ((Context.track_step, Context.track_new), (O.track_step, O.track_new)):
        mk = lambda box, stamp, ident=1: new(arm(*box), ident, stamp, (0.0, 0.0, 1.0))
        # empty tracking list: this frame's armours become the targets
        tr = step([], [mk((100, 100, 40, 40), 10), mk((300, 100, 40, 40), 10)])
        assert len(tr) == 2 and tr[0]["initialized"] == 0
        # no observations: nothing happens (main.cpp:61 `continue`), not even ageing
        assert step(tr, []).tobytes() == tr.tobytes()
        # one target re-observed (IoU > 0.5), one missed, one new armour
        tr2 = step(tr, [mk((104, 102, 40, 40), 20, ident=4), mk((600, 400, 30, 30), 20)])
        assert len(tr2) == 3
        assert tr2[0]["initialized"] == 1 and tr2[0]["timestamp"] == 20 and tr2[0]["lost_count"] == 0 and tr2[0]["ids"][0] == 4
        assert tr2[0]["armour"]["bbox"].tolist() == [100, 100, 40, 40]          # the box is never refreshed (reference behaviour)
        assert tr2[1]["lost_count"] == 1 and tr2[1]["timestamp"] == 10 and tr2[2]["armour"]["bbox"][0] == 600
        # IoU exactly at the 0.5 boundary is NOT a match: boxes 40x40 shifted by 40/3 px overlap by (80/3*40) / (2*1600 - 80/3*40) = 0.5
        # (not representable -> take a clear miss and a clear hit instead)
        tr3 = step(tr, [mk((115, 100, 40, 40), 30)])                             # IoU = 25*40/(3200-1000) = 0.4545
        assert len(tr3) == 3 and tr3[0]["lost_count"] == 1 and tr3[2]["armour"]["bbox"][0] == 115
        # ageing: a target is dropped on its 27th miss, and the target BEHIND it in the list is skipped in that pass
        tr = step([], [mk((100, 100, 40, 40), 0), mk((300, 100, 40, 40), 0), mk((500, 100, 40, 40), 0)])
        far = lambda s: [mk((900, 700, 20, 20), s)]
        for k in range(26):
            tr = step(tr, far(k + 1))
            tr = tr[:3]                                                           # drop the far armour again: keep the scenario small
            assert [int(t["lost_count"]) for t in tr] == [k + 1] * 3
        tr = step(tr, far(100))
        # pass 27: target 0 has lost_count 26 > 25 -> erased; old target 1 moved to slot 0 and is skipped (not aged); old target 2, now
        # in slot 1, is examined with i = 1: 26 > 25 -> erased too.  Left: the skipped one + the far armour
        assert len(tr) == 2 and tr[0]["armour"]["bbox"][0] == 300 and tr[0]["lost_count"] == 26 and tr[1]["armour"]["bbox"][0] == 900


def test_association_loop_product_equals_oracle_on_random_traffic():
    rng = np.random.default_rng(21)
    ta, tb = [], []
    stamp = 0
    for frame in range(120):
        stamp += 8_000_000
        obs_a, obs_b = [], []
        for _ in range(int(rng.integers(0, 5))):
            box = (float(rng.integers(0, 6)) * 120 + float(rng.integers(-12, 13)), 200 + float(rng.integers(-10, 11)), 60.0, 50.0)
            pos, ident = rng.normal(size=3), int(rng.integers(0, 7))
            obs_a.append(Context.track_new(arm(*box), ident, stamp, pos))
            obs_b.append(O.track_new(arm(*box), ident, stamp, pos))
        ta, tb = Context.track_step(ta, obs_a), O.track_step(tb, obs_b)
        assert len(ta) == len(tb) and (len(ta) == 0 or ta.tobytes() == tb.tobytes()), frame
    assert len(ta) > 0 and max(int(t["initialized"]) for t in ta) == 1


def test_association_loop_has_no_observation_limit():
    """the reference's vectors are unbounded (executable/main.cpp:69-84); round 2 stopped at 64 observations"""
    rng = np.random.default_rng(5)
    mk = lambda new, k, stamp: new(arm(40.0 * (k % 40), 60.0 * (k // 40), 30.0, 30.0), k % 7, stamp, (float(k), 0.0, 1.0))
    ta = Context.track_step([], [mk(Context.track_new, k, 1000) for k in range(150)], cap=512)
    tb = O.track_step([], [mk(O.track_new, k, 1000) for k in range(150)], cap=512)
    assert len(ta) == 150 and ta.tobytes() == tb.tobytes()
    order = rng.permutation(200)                         # 150 re-observed in another order + 50 new ones, all in one frame
    ta = Context.track_step(ta, [mk(Context.track_new, int(k), 2000) for k in order], cap=512)
    tb = O.track_step(tb, [mk(O.track_new, int(k), 2000) for k in order], cap=512)
    assert len(ta) == 200 and ta.tobytes() == tb.tobytes()
    assert all(int(t["initialized"]) == 1 and int(t["timestamp"]) == 2000 for t in ta[:150])


def test_association_loop_capacity_error_changes_nothing():
    mk = lambda k, stamp: Context.track_new(arm(50.0 * k, 10.0, 30.0, 30.0), 1, stamp, (0.0, 0.0, 1.0))
    tr = Context.track_step([], [mk(k, 10) for k in range(6)], cap=8)
    before = tr.tobytes()
    import ctypes as C
    from rmcv_amd import abi
    buf = np.zeros(8, abi.TRACK)
    buf[:6] = tr
    obs = np.array([mk(k, 20) for k in (0, 1, 20, 21, 22)], abi.TRACK)   # 6 + 5 > 8: refused before the first association
    obs_before = obs.tobytes()
    nt, no = C.c_int32(6), C.c_int32(5)
    rc = abi.lib().rmcv_track_step(buf.ctypes.data_as(C.c_void_p), C.byref(nt), 8, obs.ctypes.data_as(C.c_void_p), C.byref(no), C.c_double(1e9))
    assert rc == abi.ERR_CAPACITY and nt.value == 6 and no.value == 5
    assert buf[:6].tobytes() == before and obs.tobytes() == obs_before
    with pytest.raises(RmcvError):
        Context.track_step(tr, [mk(k, 20) for k in (0, 1, 20, 21, 22)], cap=8)


def test_association_loop_steady_state_needs_no_spare_capacity():
    """N tracked targets re-observed by N matching observations leave N entries (executable/main.cpp:69-83): cap = N is enough
    (round 3 asked for n_tracking + n_obs <= cap up front, i.e. 2 N)"""
    mk = lambda new, k, stamp: new(arm(50.0 * k, 10.0, 30.0, 30.0), 1, stamp, (float(k), 0.0, 1.0))
    ta = Context.track_step([], [mk(Context.track_new, k, 10) for k in range(6)], cap=6)
    tb = O.track_step([], [mk(O.track_new, k, 10) for k in range(6)], cap=16)
    for stamp in (20, 30, 40):
        ta = Context.track_step(ta, [mk(Context.track_new, k, stamp) for k in (3, 1, 0, 5, 4, 2)], cap=6)
        tb = O.track_step(tb, [mk(O.track_new, k, stamp) for k in (3, 1, 0, 5, 4, 2)], cap=16)
        assert len(ta) == 6 and ta.tobytes() == tb.tobytes()
    with pytest.raises(RmcvError):                                  # one observation that matches nothing: 7 entries do not fit 6
        Context.track_step(ta, [mk(Context.track_new, k, 50) for k in (0, 1, 2, 3, 4, 9)], cap=6)
