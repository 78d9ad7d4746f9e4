"""CPU tests of the pose oracle (oracle/rmcv_oracle_pnp.c): rm::solve_PnP = undistortPoints + IPPE for a square.
The reference has no tests.  The checks are this build's own: poses chosen here are projected through the
camera model (distortion included) with numpy in float64 and the solver has to find them again."""
import numpy as np
import pytest

import oracle_lib as O


def rodrigues(rvec):
    th = np.linalg.norm(rvec)
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def project(R, t, cfg, half=13.5, distort=True):
    """armour.vertices for a square of side 2*half with pose (R, t): image points of the object corners in the order the
    reference feeds them (mobility.cpp:175-185: object corners 0..3 <-> vertices 1, 2, 3, 0)"""
    K = np.array(cfg.camera_matrix).reshape(3, 3)
    k1, k2, p1, p2, k3 = list(cfg.dist)
    obj = np.array([[-half, half, 0], [half, half, 0], [half, -half, 0], [-half, -half, 0]], float)
    cam = obj @ R.T + t
    x, y = cam[:, 0] / cam[:, 2], cam[:, 1] / cam[:, 2]
    if distort:
        r2 = x * x + y * y
        rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
        xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = xd, yd
    img = np.stack([K[0, 0] * x + K[0, 2], K[1, 1] * y + K[1, 2]], 1)
    v = np.zeros((4, 2), np.float32)
    for i, k in enumerate((1, 2, 3, 0)):
        v[k] = img[i]
    return v


def test_default_config_is_the_float_literals():
    c = O.default_pnp_config()
    assert c.camera_matrix[0] == float(np.float32(1782.672144409928)) and c.camera_matrix[8] == 1.0
    assert c.dist[4] == float(np.float32(-0.3181808766352414))
    assert c.gripper2camera[3] == float(np.float32(-27.25811584661768)) and c.gripper2camera[15] == 1.0
    assert (c.square_w, c.square_h) == (27.0, 27.0)


@pytest.mark.parametrize("seed", range(40))
def test_recovers_a_known_pose(seed):
    rng = np.random.default_rng(seed)
    cfg = O.default_pnp_config()
    rvec = rng.uniform(-0.7, 0.7, 3)
    R = rodrigues(rvec)
    t = np.array([rng.uniform(-300, 300), rng.uniform(-250, 250), rng.uniform(800, 4000)])
    v = project(R, t, cfg)
    rc, r, tt = O.solve_pnp(v, cfg)
    assert rc == 0
    # float32 image points (1/8 px at 1000) and the 5-step undistortion bound the accuracy: the image of the recovered pose
    # has to land on the given pixels
    back = project(rodrigues(r), tt, cfg)
    assert np.abs(back - v).max() < 0.05
    # depth within 1.5 %, lateral offset consistent with it
    assert abs(tt[2] - t[2]) / t[2] < 0.015
    assert np.linalg.norm(tt[:2] / tt[2] - t[:2] / t[2]) < 2e-3


def test_frontal_square_without_distortion_is_exact_enough():
    cfg = O.default_pnp_config()
    for i in range(5):
        cfg.dist[i] = 0.0
    t = np.array([40.0, -25.0, 1500.0])
    v = project(np.eye(3), t, cfg, distort=False)
    rc, r, tt = O.solve_pnp(v, cfg)
    assert rc == 0 and np.allclose(tt, t, rtol=2e-4) and np.linalg.norm(r) < 0.05


def test_degenerate_points():
    rc, r, t = O.solve_pnp(np.zeros((4, 2), np.float32))
    assert rc == 1 and not r.any() and not t.any()


def test_position_transform():
    cfg = O.default_pnp_config()
    G = np.array(cfg.gripper2camera).reshape(4, 4)
    rng = np.random.default_rng(1)
    B = np.eye(4)
    B[:3, :3] = rodrigues(rng.uniform(-1, 1, 3))
    arm = np.zeros(3, O.ARMOUR)
    for k in range(3):
        arm[k]["vertices"] = project(rodrigues(rng.uniform(-0.5, 0.5, 3)), np.array([10.0 * k, 5.0, 1200.0 + 300 * k]), cfg)
    r, t, p = O.locate_armours(arm, cfg, B)
    for k in range(3):
        rc, r1, t1 = O.solve_pnp(arm[k]["vertices"], cfg)
        assert np.array_equal(r[k], r1) and np.array_equal(t[k], t1)
        want = B @ (G @ np.append(t1, 1.0))
        assert np.allclose(p[k], want[:3], rtol=1e-13, atol=1e-10)
    _, _, p0 = O.locate_armours(arm, cfg, None)
    assert np.allclose(p0[0], (G @ np.append(t[0], 1.0))[:3], rtol=1e-13, atol=1e-10)


def test_math_modes_agree():
    cfg = O.default_pnp_config()
    v = project(rodrigues(np.array([0.3, -0.2, 0.1])), np.array([50.0, 20.0, 2000.0]), cfg)
    O.set_math_mode(1)
    _, r1, t1 = O.solve_pnp(v, cfg)
    O.set_math_mode(0)
    _, r0, t0 = O.solve_pnp(v, cfg)
    assert np.array_equal(t0, t1) and np.allclose(r0, r1, rtol=1e-14, atol=1e-15)
