"""GPU parity of the pose row (rm::solve_PnP + the world transform of executable/main.cpp:183-192), through
the C-ABI, bit-for-bit (fp64) against oracle/rmcv_oracle_pnp.c."""
import numpy as np
import pytest

from rmcv_amd import CAMP_BLUE, STAGE_ALL, STAGE_POSE, default_params, default_pnp_config, synth
from test_oracle_pnp import project, rodrigues

pytestmark = pytest.mark.gpu


def to_oracle_cfg(oracle, cfg):
    o = oracle.PnpConfig()
    for name in ("camera_matrix", "dist", "gripper2camera"):
        for i, v in enumerate(getattr(cfg, name)):
            getattr(o, name)[i] = v
    o.square_w, o.square_h = cfg.square_w, cfg.square_h
    return o


def test_default_config_matches_oracle(oracle):
    a, b = default_pnp_config(), oracle.default_pnp_config()
    assert bytes(a) == bytes(b)


def test_locate_armours_random_poses(ctx, oracle):
    rng = np.random.default_rng(5)
    cfg = default_pnp_config()
    ctx.pnp_load(cfg)
    ocfg = to_oracle_cfg(oracle, cfg)
    n = 200
    arm = np.zeros(n, oracle.ARMOUR)
    for k in range(n):
        R = rodrigues(rng.uniform(-0.9, 0.9, 3))
        t = np.array([rng.uniform(-400, 400), rng.uniform(-300, 300), rng.uniform(500, 6000)])
        arm[k]["vertices"] = project(R, t, ocfg)
    arm[7]["vertices"] = 0                                    # degenerate: all four points coincide
    arm[8]["vertices"] = arm[9]["vertices"][[0, 0, 2, 2]]     # degenerate: two pairs of coincident points
    B = np.eye(4)
    B[:3, :3] = rodrigues(np.array([0.2, -0.4, 0.1]))
    B[:3, 3] = (5.0, -7.0, 11.0)
    for base in (None, B):
        got = ctx.locate_armours(arm, base)
        want = oracle.locate_armours(arm, ocfg, base)
        for g, w, name in zip(got, want, ("rvec", "tvec", "position")):
            assert g.tobytes() == w.tobytes(), (name, np.abs(g - w).max())


def test_locate_armours_other_camera(ctx, oracle):
    cfg = default_pnp_config()
    cfg.camera_matrix[0], cfg.camera_matrix[4], cfg.camera_matrix[2], cfg.camera_matrix[5] = 1200.0, 1210.0, 640.0, 512.0
    for i, v in enumerate((0.08, -0.2, 0.001, -0.002, 0.05)):
        cfg.dist[i] = v
    cfg.square_w, cfg.square_h = 13.5, 5.5                    # IPPE_SQUARE is fed a rectangle when exactSize is one (mobility.cpp:175)
    ctx.pnp_load(cfg)
    ocfg = to_oracle_cfg(oracle, cfg)
    rng = np.random.default_rng(6)
    arm = np.zeros(64, oracle.ARMOUR)
    arm["vertices"] = rng.uniform(100, 900, (64, 4, 2)).astype(np.float32)   # arbitrary quadrilaterals
    got, want = ctx.locate_armours(arm), oracle.locate_armours(arm, ocfg)
    for g, w in zip(got, want):
        assert g.tobytes() == w.tobytes()
    ctx.pnp_load(default_pnp_config())


def test_batch_pose_stage(ctx, oracle):
    """the batch path with RMCV_STAGE_POSE: every armour of every frame, per-frame base2gripper"""
    n = 6
    frames = synth.batch(900, n, 1280, 1024, CAMP_BLUE, 0)
    ctx.pnp_load()
    rng = np.random.default_rng(2)
    mats = np.tile(np.eye(4), (n, 1, 1))
    for f in range(n):
        mats[f, :3, :3] = rodrigues(rng.uniform(-1, 1, 3))
        mats[f, :3, 3] = rng.uniform(-50, 50, 3)
    ctx.upload(frames)
    ctx.set_base2gripper(mats)
    ctx.run(default_params(), STAGE_ALL | STAGE_POSE)
    ctx.sync()
    arm, offs = ctx.armours()
    r, t, p = ctx.poses()
    assert len(r) == len(arm) > 0
    ocfg = oracle.default_pnp_config()
    for f in range(n):
        a = arm[offs[f]:offs[f + 1]]
        wr, wt, wp = oracle.locate_armours(a, ocfg, mats[f])
        assert r[offs[f]:offs[f + 1]].tobytes() == wr.tobytes() and t[offs[f]:offs[f + 1]].tobytes() == wt.tobytes(), f
        assert p[offs[f]:offs[f + 1]].tobytes() == wp.tobytes(), f
    # the synthetic armours are a few metres away in the default camera
    assert np.all(t[:, 2] > 100) and np.all(t[:, 2] < 1e5)


def test_pose_stage_needs_config():
    from rmcv_amd import Context, RmcvError
    c = Context(device=0, max_frames=1, max_width=256, max_height=256)
    c.upload(np.zeros((1, 64, 64, 3), np.uint8))
    with pytest.raises(RmcvError):
        c.run(default_params(), STAGE_ALL | STAGE_POSE)
    c.close()
