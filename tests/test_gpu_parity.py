"""GPU parity proper: every stage of the HIP path, through the C-ABI, bit-for-bit against the CPU
oracle on the same seeded inputs.  (The oracle restates the reference; see oracle/rmcv_oracle.h
for its own pinning status.)"""
import numpy as np
import pytest

from rmcv_amd import (CAMP_BLUE, CAMP_GUIDELIGHT, CAMP_NEUTRAL, CAMP_RED, MORPH_CLOSE, MORPH_DILATE, MORPH_NONE,
                      STAGE_ALL, STAGE_BINARY, default_params, synth)

pytestmark = pytest.mark.gpu


def rand_bgr(rng, h, w):
    """noise with structured bright patches so the threshold fires on blobs of every shape"""
    img = rng.integers(0, 48, (h, w, 3), dtype=np.uint8)
    for _ in range(12):
        y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
        hh, ww = int(rng.integers(1, max(2, h // 3))), int(rng.integers(1, max(2, w // 4)))
        img[y:y + hh, x:x + ww] = (255, 180, 20)
    m = rng.random((h, w)) < 0.02
    img[m] = (255, 200, 0)
    m = rng.random((h, w)) < 0.03
    img[m] = (0, 0, 0)
    return img


@pytest.mark.parametrize("shape", [(64, 64), (65, 130), (37, 200), (128, 256), (100, 1), (1, 77), (200, 333)])
@pytest.mark.parametrize("morph", [MORPH_NONE, MORPH_DILATE, MORPH_CLOSE])
def test_binary_small(ctx, oracle, shape, morph):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1] + morph)
    img = rand_bgr(rng, *shape)
    for camp, lb in [(CAMP_BLUE, 80), (CAMP_RED, 80), (CAMP_GUIDELIGHT, 100), (CAMP_NEUTRAL, 1), (CAMP_BLUE, 0), (CAMP_BLUE, 256)]:
        _, _, binary = ctx.extract_color_csr(img, camp, lb, morph)
        ref = oracle.extract_binary(img, camp, lb, morph)
        assert np.array_equal(binary, ref), (shape, morph, camp, lb)


@pytest.mark.parametrize("seed", range(6))
def test_contours_random(ctx, oracle, seed):
    rng = np.random.default_rng(100 + seed)
    h, w = [(96, 160), (130, 70), (64, 64), (200, 320), (77, 129), (256, 256)][seed]
    img = rand_bgr(rng, h, w)
    pts, offs, binary = ctx.extract_color_csr(img, CAMP_BLUE, 80, MORPH_CLOSE)
    rb = oracle.extract_binary(img, CAMP_BLUE, 80, MORPH_CLOSE)
    rp, ro = oracle.find_contours(rb)
    assert np.array_equal(binary, rb)
    assert np.array_equal(offs, ro)
    assert np.array_equal(pts, rp)


@pytest.mark.parametrize("variant", [0, 1])
def test_full_path_synthetic(ctx, oracle, variant):
    """BASELINE config 3 at test size: every stage output identical on the synthetic stream"""
    n = 8
    frames = synth.batch(variant * 1000, n, 1280, 1024, CAMP_BLUE, variant)
    p = default_params()
    ctx.upload(frames)
    ctx.run(p, STAGE_ALL)
    ctx.sync()
    arm, aoffs = ctx.armours()
    cnt = ctx.counts()
    assert not cnt["status"].any()            # the synthetic streams stay on the LDS tables: neither the mid tier nor the sequential scanner
    for f in range(n):
        ref = oracle.detect_frame(frames[f], oracle.default_params())
        assert np.array_equal(ctx.binary(f), ref["binary"]), f
        pts, offs = ctx.contours(f)
        assert np.array_equal(offs, ref["offs"]), f
        assert np.array_equal(pts, ref["pts"]), f
        blobs, _ = ctx.blobs(f)
        assert blobs.tobytes() == ref["blobs"].tobytes(), f
        a = arm[aoffs[f]:aoffs[f + 1]]
        assert a["vertices"].tobytes() == ref["armours"]["vertices"].tobytes(), f
        assert a.tobytes() == ref["armours"].tobytes(), f


# ---------------------------------------------------------------- stage-wise C-ABI entry points
def test_stagewise_lightblobs_and_armours(ctx, oracle):
    for idx in (11, 12, 1013):
        f = synth.frame(idx, 1280, 1024, CAMP_BLUE, 1 if idx > 1000 else 0)
        ref = oracle.detect_frame(f)
        blobs, src, neg = ctx.filter_lightblobs(ref["pts"], ref["offs"])
        rb, rs, rn = oracle.filter_lightblobs(ref["pts"], ref["offs"])
        assert blobs.tobytes() == rb.tobytes() and np.array_equal(src, rs) and np.array_equal(neg, rn)
        arm = ctx.filter_armours(rb)
        assert arm.tobytes() == oracle.filter_armours(rb).tobytes()
    assert len(ctx.filter_armours(np.zeros(0, blobs.dtype))) == 0
    b, s, n = ctx.filter_lightblobs(np.zeros(0, ref["pts"].dtype), np.zeros(1, np.int32))
    assert len(b) == 0 and len(n) == 0


def test_fit_ellipse_both_paths(ctx, oracle):
    import math
    from test_oracle_kat import ellipse_points
    for (A, B, n) in [(400, 150, 720), (36, 12, 40), (90, 14, 230), (12, 5, 24)]:
        for deg in (0, 17, 60, 90, 133):
            p = ellipse_points(700, 600, A, B, math.radians(deg), n)
            r, _ = oracle.fit_ellipse_direct(p)
            g = ctx.fit_ellipse(p)
            assert g.tobytes() == r.tobytes(), (A, B, n, deg, g, r)


def test_pairing_kats_on_gpu(ctx, oracle):
    from test_oracle_kat import blob
    sets = [[blob(100, 100, 40), blob(200, 100, 40)], [blob(100, 100, 40), blob(170, 100, 40), blob(240, 100, 40)],
            [blob(100, 100, 40), blob(130, 118, 40)], [blob(100, 100, 40), blob(200, 100, 12)],
            [blob(100 + 37 * k, 100 + (k % 3) * 5, 30 + k) for k in range(70)]]
    for s in sets:
        arr = np.array(s, oracle.LIGHTBLOB)
        for kw in ({}, dict(shear_max=40.0), dict(length_ratio_max=0.2), dict(angle_diff_max=999.0, shear_max=999.0, length_ratio_max=0.01)):
            p = oracle.default_params(**kw)
            ref = oracle.filter_armours(arr, p)
            if len(ref) > ctx.limits.max_armours:
                continue
            got = ctx.filter_armours(arr, p.angle_diff_max, p.shear_max, p.length_ratio_max)
            assert got.tobytes() == ref.tobytes()


# ---------------------------------------------------------------- nested components: the external rule
def test_contours_nested_shapes(ctx, oracle):
    h, w = 96, 192
    canvas = np.zeros((h, w), np.uint8)
    canvas[4:40, 4:60] = 255
    canvas[10:34, 10:54] = 0            # ring
    canvas[16:28, 20:44] = 255          # island in the hole
    canvas[19:25, 26:38] = 0            # with its own hole
    canvas[21:23, 30:33] = 255          # and an island in that
    canvas[50:90, 70:130] = 255
    canvas[50:80, 80:120] = 0           # U cavity
    canvas[60:70, 90:100] = 255         # blob in the cavity: kept
    canvas[5:90, 150] = 255             # 1-px vertical line
    canvas[45, 140:190] = 255           # crossing 1-px horizontal line
    canvas[0, 0] = canvas[h - 1, w - 1] = canvas[0, w - 1] = 255
    rng = np.random.default_rng(5)
    for trial in range(4):
        img = np.zeros((h, w, 3), np.uint8)
        c = canvas.copy()
        if trial:
            c[rng.random((h, w)) < 0.01 * trial] ^= 255
        img[..., 0] = c
        pts, offs, binary = ctx.extract_color_csr(img, CAMP_BLUE, 80, MORPH_NONE)
        assert np.array_equal(binary, c)
        rp, ro = oracle.find_contours(c)
        assert np.array_equal(offs, ro) and np.array_equal(pts, rp), trial


@pytest.mark.parametrize("density", [0.02, 0.2, 0.5, 0.8])
def test_contours_noise(ctx, oracle, density):
    """salt-and-pepper at several densities: thousands of tiny, touching and nested components"""
    rng = np.random.default_rng(int(density * 100))
    h, w = 120, 200
    c = ((rng.random((h, w)) < density) * 255).astype(np.uint8)
    img = np.zeros((h, w, 3), np.uint8)
    img[..., 0] = c
    pts, offs, binary = ctx.extract_color_csr(img, CAMP_BLUE, 80, MORPH_NONE)
    rp, ro = oracle.find_contours(c)
    assert np.array_equal(offs, ro) and np.array_equal(pts, rp)


# ---------------------------------------------------------------- committed golden vectors
def test_golden_vectors_gpu(ctx):
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synthetic_armours.json")))
    frames = np.stack([synth.frame(r["index"], g["width"], g["height"], g["camp"], r["variant"]) for r in g["frames"]])
    arm, offs = ctx.detect_batch(frames)
    cnt = ctx.counts()
    for k, rec in enumerate(g["frames"]):
        a = arm[offs[k]:offs[k + 1]]
        got = [[float.hex(float(v)) for v in x["vertices"].reshape(-1)] for x in a]
        assert got == rec["armour_vertices_hex"], rec["index"]
        assert cnt["n_contours"][k] == rec["n_contours"] and cnt["n_points"][k] == rec["n_points"]
        assert cnt["n_blobs"][k] == rec["n_blobs"]


# ---------------------------------------------------------------- capacity errors and geometry changes
def test_capacity_is_reported(oracle):
    from rmcv_amd import Context, RmcvError
    small = Context(device=0, max_frames=1, max_width=256, max_height=128, max_contours=4, max_points=64, max_blobs=2,
                    max_armours=1)
    img = np.zeros((128, 256, 3), np.uint8)
    for k in range(8):
        img[10:60, 10 + 30 * k:16 + 30 * k, 0] = 255
    with pytest.raises(RmcvError) as e:
        small.extract_color_csr(img, CAMP_BLUE, 80)
    assert e.value.code == -2
    with pytest.raises(RmcvError):
        small.extract_color_csr(np.zeros((200, 300, 3), np.uint8), CAMP_BLUE, 80)   # larger than the context
    rb = np.array([oracle.make_lightblob(np.array((100 + 70 * k, 100, 10, 40, 0), oracle.RRECT), 1) for k in range(2)])
    assert len(small.filter_armours(rb)) == 1
    rb3 = np.array([oracle.make_lightblob(np.array((100 + 70 * k, 100, 10, 40, 0), oracle.RRECT), 1) for k in range(3)])
    with pytest.raises(RmcvError) as e:
        small.filter_armours(rb3)
    assert e.value.code == -2
    small.close()


def test_geometry_changes_on_one_context(ctx, oracle):
    rng = np.random.default_rng(9)
    for (h, w) in [(200, 333), (64, 64), (130, 640), (64, 64), (1024, 1280), (90, 100)]:
        img = rand_bgr(rng, h, w) if h < 1000 else synth.frame(77, w, h, CAMP_BLUE, 1)
        pts, offs, binary = ctx.extract_color_csr(img, CAMP_BLUE, 80, MORPH_CLOSE)
        rb = oracle.extract_binary(img, CAMP_BLUE, 80, MORPH_CLOSE)
        rp, ro = oracle.find_contours(rb)
        assert np.array_equal(binary, rb) and np.array_equal(offs, ro) and np.array_equal(pts, rp), (h, w)


# ---------------------------------------------------------------- BASELINE full size: 256 x 1280x1024
def test_full_size_batch_properties(oracle):
    """size-independent properties at BASELINE.json's full batch, plus the oracle on every frame's armour list"""
    from rmcv_amd import Context
    n = 256
    big = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    frames = synth.batch(0, n, 1280, 1024, CAMP_BLUE, 0, threads=16)
    arm, offs = big.detect_batch(frames)
    cnt = big.counts()
    assert not (cnt["status"] & 15).any()
    bin7 = big.binary(7)
    # (1) determinism: a second run gives the same bytes
    arm2, offs2 = big.detect_batch(frames)
    assert arm.tobytes() == arm2.tobytes() and np.array_equal(offs, offs2)
    # (2) frames are independent: two half batches == the full batch (the sharding property of config 4)
    a0, o0 = big.detect_batch(frames[:n // 2])
    a1, o1 = big.detect_batch(frames[n // 2:])
    assert a0.tobytes() + a1.tobytes() == arm.tobytes()
    assert np.array_equal(np.concatenate([o0, o1[1:] + o0[-1]]), offs)
    # (3) closing is idempotent: feed a closed binary back in as the blue channel
    fb = np.zeros((2, 1024, 1280, 3), np.uint8)
    fb[0, ..., 0] = bin7
    fb[1, ..., 0] = bin7
    big.upload(fb)
    big.run(default_params(), STAGE_BINARY)
    big.sync()
    assert np.array_equal(big.binary(0), bin7) and np.array_equal(big.binary(1), bin7)
    # (4) the reference path (oracle) on every frame: the emitted armour vertex lists are identical
    p = oracle.default_params()
    for f in range(n):
        ref = oracle.detect_frame(frames[f], p)["armours"]
        assert arm[offs[f]:offs[f + 1]].tobytes() == ref.tobytes(), f
    assert offs[-1] > n
    big.close()


# ---------------------------------------------------------------- limits of the fast contour path -> literal scanner
def _check_contours(c, canvas, oracle):
    img = np.zeros(canvas.shape + (3,), np.uint8)
    img[..., 0] = canvas
    pts, offs, binary = c.extract_color_csr(img, CAMP_BLUE, 80, MORPH_NONE)
    rp, ro = oracle.find_contours(canvas)
    assert np.array_equal(binary, canvas) and np.array_equal(offs, ro) and np.array_equal(pts, rp)
    return len(ro) - 1, len(rp)


def test_contours_beyond_fast_path_limits(oracle):
    from rmcv_amd import Context
    c = Context(device=0, max_frames=1, max_width=2304, max_height=2200, max_contours=8192, max_points=1 << 18)
    # (1) one contour longer than the 2048 step codes a wavefront records: a serpentine
    a = np.zeros((400, 600), np.uint8)
    for k in range(0, 380, 8):
        a[k:k + 4, 10:590] = 255
        a[k + 4:k + 8, (10 if (k // 8) % 2 else 586):(14 if (k // 8) % 2 else 590)] = 255
    nc, npnt = _check_contours(c, a, oracle)
    assert nc == 1 and npnt > 2048
    # (2) a solid frame-filling rectangle: perimeter 2 * (1500 + 900)
    b = np.zeros((1000, 1600), np.uint8)
    b[50:950, 50:1550] = 255
    assert _check_contours(c, b, oracle) == (1, 2 * (1500 + 900) - 4)
    # (3) more local tops than the candidate queue holds (> 2048 components)
    d = np.zeros((600, 800), np.uint8)
    d[::8, ::10] = 255
    nc, _ = _check_contours(c, d, oracle)
    assert nc == 75 * 80
    # (4) taller than the LDS row tables (> 2048 rows) and (5) wider than 32 words (> 2048 px)
    e = np.zeros((2100, 300), np.uint8)
    e[5:2090:40, 10:250] = 255
    e[3:2095, 280] = 255
    _check_contours(c, e, oracle)
    g = np.zeros((120, 2300), np.uint8)
    g[10:100:9, 5:2290] = 255
    g[10:100, 2295] = 255
    _check_contours(c, g, oracle)
    # (6) more non-empty words than the LDS label store holds (> 2048): vertical stripes over 280 rows x 20 words
    t = np.zeros((300, 1280), np.uint8)
    t[10:290, ::8] = 255
    nc, _ = _check_contours(c, t, oracle)
    assert nc == 160
    c.close()


def test_borrowed_device_frames_with_row_and_frame_padding(oracle):
    """rmcv_batch_set_device_frames on HBM the caller owns (a torch tensor), with a row pitch and a frame pitch that are not
    the packed 3*w / 3*w*h: the byte-wise loader of k_binary and the fast one must both honour them"""
    import torch
    from rmcv_amd import Context
    dev = torch.device("cuda", 0)
    for (h, w, stride, extra) in [(96, 200, 3 * 200 + 7, 5), (64, 128, 3 * 128 + 16, 64), (1024, 1280, 3 * 1280, 4096)]:
        n = 3
        frames = np.stack([synth.frame(20 + i, w, h) if w >= 64 and h >= 64 else None for i in range(n)])
        pitch = stride * h + extra
        buf = torch.zeros(n * pitch + 64, dtype=torch.uint8, device=dev)
        host = np.zeros(n * pitch, np.uint8)
        for f in range(n):
            for y in range(h):
                host[f * pitch + y * stride:f * pitch + y * stride + 3 * w] = frames[f, y].reshape(-1)
        buf[:n * pitch] = torch.from_numpy(host).to(dev)
        c = Context(device=0, max_frames=n, max_width=w, max_height=h)
        c.bind_device_frames(buf.data_ptr(), n, h, w, stride=stride, frame_pitch=pitch, keepalive=buf)
        c.run(default_params(), STAGE_ALL)
        c.sync()
        arm, offs = c.armours()
        for f in range(n):
            ref = oracle.detect_frame(frames[f])
            assert np.array_equal(c.binary(f), ref["binary"]), (h, w, f)
            assert arm[offs[f]:offs[f + 1]].tobytes() == ref["armours"].tobytes(), (h, w, f)
        c.close()


def test_sparse_kernel_with_four_wavefronts(oracle):
    """RMCV_OPT_SPARSE_WAVES = 4 (the throughput setting bench.py uses with several batches in flight): same results"""
    from rmcv_amd import OPT_SPARSE_WAVES, Context, RmcvError
    n = 6
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    with pytest.raises(RmcvError):
        c.set_option(OPT_SPARSE_WAVES, 3)
    with pytest.raises(RmcvError):
        c.set_option(99, 1)
    c.set_option(OPT_SPARSE_WAVES, 4)
    for variant in (0, 1):
        frames = synth.batch(4000 + variant, n, 1280, 1024, CAMP_BLUE, variant)
        c.upload(frames)
        c.run(default_params(), STAGE_ALL)
        c.sync()
        arm, aoffs = c.armours()
        for f in range(n):
            ref = oracle.detect_frame(frames[f], oracle.default_params())
            pts, offs = c.contours(f)
            assert np.array_equal(offs, ref["offs"]) and np.array_equal(pts, ref["pts"]), f
            blobs, _ = c.blobs(f)
            assert blobs.tobytes() == ref["blobs"].tobytes(), f
            assert arm[aoffs[f]:aoffs[f + 1]].tobytes() == ref["armours"].tobytes(), f
    # nested shapes (a blob inside a hole is dropped by RETR_EXTERNAL) stay on the LDS tables; 23 1-px lines of 100 pixels (more
    # than 4096 border visits) take the mid tier inside the 4-wavefront workgroup
    img = np.zeros((2, 256, 256, 3), np.uint8)
    img[0, 40:200, 40:200] = (255, 0, 0)
    img[0, 80:160, 80:160] = 0
    img[0, 100:140, 100:140] = (255, 0, 0)
    for r in range(23):
        img[1, 4 + 3 * r, 10:110] = (255, 0, 0)
    c.upload(img)
    p_none = default_params()
    p_none.morph = MORPH_NONE
    c.run(p_none, STAGE_ALL)
    c.sync()
    for f in range(2):
        rb = oracle.extract_binary(img[f], CAMP_BLUE, 80, MORPH_NONE)
        rp, ro = oracle.find_contours(rb)
        pts, offs = c.contours(f)
        assert np.array_equal(offs, ro) and np.array_equal(pts, rp), f
    st = c.counts()["status"]
    assert st[0] == 0 and st[1] == 64 and len(c.contours(0)[1]) - 1 == 1
    c.close()


@pytest.mark.parametrize("variant,waves", [(0, 4), (1, 8)])
def test_full_size_every_stage_of_every_frame(oracle, variant, waves):
    """BASELINE.json's full batch (256 x 1280x1024), both synthetic streams, both sparse-kernel settings: binary image, contours,
    light blobs and armours of EVERY frame against the oracle (run on 16 host threads)"""
    from concurrent.futures import ThreadPoolExecutor

    from rmcv_amd import OPT_SPARSE_WAVES, Context
    n = 256
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    c.set_option(OPT_SPARSE_WAVES, waves)
    frames = synth.batch(50000 + 1000 * variant, n, 1280, 1024, CAMP_BLUE, variant, threads=16)
    arm, offs = c.detect_batch(frames)
    assert not (c.counts()["status"] & 15).any()
    p = oracle.default_params()
    with ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(lambda f: oracle.detect_frame(frames[f], p), range(n)))
    n_blobs = 0
    for f in range(n):
        ref = refs[f]
        assert np.array_equal(c.binary(f), ref["binary"]), f
        pts, co = c.contours(f)
        assert np.array_equal(co, ref["offs"]) and np.array_equal(pts, ref["pts"]), f
        blobs, _ = c.blobs(f)
        assert blobs.tobytes() == ref["blobs"].tobytes(), f
        assert arm[offs[f]:offs[f + 1]].tobytes() == ref["armours"].tobytes(), f
        n_blobs += len(blobs)
    assert n_blobs > 2 * n and offs[-1] > n // 2
    c.close()


def test_no_image_modifier_changes_nothing_but_the_image(ctx, oracle):
    """RMCV_STAGE_NO_IMAGE: contours / blobs / armours as with the image; the byte image buffer is left alone"""
    from rmcv_amd import STAGE_NO_IMAGE
    n = 4
    frames = synth.batch(600, n, 1280, 1024, CAMP_BLUE, 1)
    ctx.upload(np.zeros_like(frames))
    ctx.run(default_params(), STAGE_ALL)            # leaves an all-zero image in the buffer
    ctx.sync()
    ctx.upload(frames)
    ctx.run(default_params(), STAGE_ALL | STAGE_NO_IMAGE)
    ctx.sync()
    arm, offs = ctx.armours()
    for f in range(n):
        ref = oracle.detect_frame(frames[f], oracle.default_params())
        pts, co = ctx.contours(f)
        assert np.array_equal(co, ref["offs"]) and np.array_equal(pts, ref["pts"]), f
        assert arm[offs[f]:offs[f + 1]].tobytes() == ref["armours"].tobytes(), f
        assert not ctx.binary(f).any()              # stale on purpose
    ctx.run(default_params(), STAGE_ALL)
    ctx.sync()
    assert np.array_equal(ctx.binary(0), oracle.detect_frame(frames[0], oracle.default_params())["binary"])


def test_c5_geometry_every_frame(oracle):
    """BASELINE config 5's geometry (1920x1200: 37.5 strips per frame, so the last strip of every frame is partial and the tapered
    strip queue hands out pieces that lie wholly below the image): binary, contours, armours of every frame"""
    from concurrent.futures import ThreadPoolExecutor

    from rmcv_amd import Context
    n = 64
    c = Context(device=0, max_frames=n, max_width=1920, max_height=1200)
    frames = synth.batch(70000, n, 1920, 1200, CAMP_BLUE, 1, threads=16)
    arm, offs = c.detect_batch(frames)
    p = oracle.default_params()
    with ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(lambda f: oracle.detect_frame(frames[f], p), range(n)))
    for f in range(n):
        assert np.array_equal(c.binary(f), refs[f]["binary"]), f
        pts, co = c.contours(f)
        assert np.array_equal(co, refs[f]["offs"]) and np.array_equal(pts, refs[f]["pts"]), f
        assert arm[offs[f]:offs[f + 1]].tobytes() == refs[f]["armours"].tobytes(), f
    c.close()


def test_cycle_formulation_paths(oracle):
    """which path findContours takes: thin lines / diagonals / spurs (pixels visited twice or -- up to 32 per frame -- three times) stay on
    the LDS tables (status 0); more such junctions, more than 4096 border visits or more than 512 contours take the mid tier
    (status bit 64: tables in global memory, never the sequential scanner); the results are equal either way"""
    from rmcv_amd import Context
    c = Context(device=0, max_frames=1, max_width=512, max_height=512, max_contours=16384, max_points=1 << 17)

    def run(canvas):
        n, npts = _check_contours(c, canvas, oracle)
        return int(c.counts()["status"][0]), n, npts

    a = np.zeros((200, 300), np.uint8)
    a[20, 30:200] = 255                                   # horizontal 1-px line: every inner pixel is visited twice
    a[40:150, 50] = 255                                   # vertical 1-px line
    for i in range(60):
        a[60 + i, 100 + i] = 255                          # diagonal
        a[60 + i, 260 - i] = 255                          # anti-diagonal
    a[160:170, 10:40] = 255
    a[165, 40:60] = 255                                   # a block with a spur
    st, n, npts = run(a)
    assert st == 0 and n == 5 and npts > 500
    b = np.zeros((64, 64), np.uint8)
    b[32, 10:50] = 255
    b[10:50, 30] = 255                                    # '+': the follower passes the centre diagonally, no pixel is visited 3 times
    st, n, _ = run(b)
    assert st == 0 and n == 1
    b = np.zeros((64, 64), np.uint8)
    for i in range(1, 12):
        b[30 - i, 30 - i] = b[30 - i, 30 + i] = b[30 + i, 30] = 255
    b[30, 30] = 255                                       # 'Y' of 1-px lines: the centre is visited three times (side table)
    st, n, _ = run(b)
    assert st == 0 and n == 1
    b = np.zeros((200, 400), np.uint8)
    for j in range(40):                                    # 40 such junctions: more than the side table lists -> literal scanner
        cx, cy = 10 + 9 * j, 100
        for i in range(1, 4):
            b[cy - i, cx - i] = b[cy - i, cx + i] = b[cy + i, cx] = 255
        b[cy, cx] = 255
    st, n, _ = run(b)
    assert st == 64 and n == 40
    d = np.zeros((300, 300), np.uint8)
    for r in range(22):
        d[4 + 3 * r, 10:110] = 255                         # 22 lines of 100 pixels = 22 x 198 = 4356 visits: more than 4096
    st, n, _ = run(d)
    assert st == 64 and n == 22
    d[4 + 3 * 20:4 + 3 * 22, :] = 0                        # 20 lines = 3960 visits fit
    st, n, npts = run(d)
    assert st == 0 and n == 20 and npts == 20 * 198
    e = np.zeros((300, 300), np.uint8)
    e[::3, ::3] = 255                                      # 10 000 isolated pixels again, but > SLOT capacity as well
    st, n, _ = run(e)
    assert st == 64 and n == 10000
    c.close()


def test_nested_components_on_the_cycle_path(oracle):
    """RETR_EXTERNAL drops every component that lies in a hole of a traced one.  The cycle path settles this by a fixed-point
    iteration on the labels (siblings inside one hole need a second round, chains of them more); results equal the oracle and the
    frames do not fall back to the literal scanner (status 0)"""
    from rmcv_amd import Context
    c = Context(device=0, max_frames=1, max_width=512, max_height=512)
    a = np.zeros((200, 260), np.uint8)                      # (about 3000 border visits: below the 4096 the cycle path holds)
    a[10:190, 10:250] = 255
    a[25:175, 25:235] = 0                                   # a big ring ...
    for k in range(5):
        a[35:55, 35 + 40 * k:55 + 40 * k] = 255             # ... five siblings in a row inside its hole (each sees its left neighbour's labels)
    a[80:160, 50:200] = 255
    a[95:145, 65:185] = 0                                   # a ring inside the hole ...
    a[110:130, 80:100] = 255                                # ... with something inside ITS hole (depth 3)
    a[2:7, 5:250:27] = 255                                  # top-level specks above the ring
    a[193:198, 10:60] = 255                                 # and a top-level bar below
    n, npts = _check_contours(c, a, oracle)
    assert int(c.counts()["status"][0]) == 0
    assert n == len(range(5, 250, 27)) + 2                  # the specks, the big ring, the bar: nothing from inside
    # a U-shaped cavity is not a hole: the blob in it is kept
    b = np.zeros((200, 200), np.uint8)
    b[20:180, 20:180] = 255
    b[20:150, 50:150] = 0
    b[60:100, 80:120] = 255
    n, _ = _check_contours(c, b, oracle)
    assert int(c.counts()["status"][0]) == 0 and n == 2
    c.close()


def test_contours_fuzz_random_scenes(oracle):
    """300 random compositions of bars, discs, rings, elliptic rings, thin lines, salt and pepper (tools/fuzz_contours.py): nested
    and touching shapes of every kind; contours equal the oracle, and most scenes stay on the cycle path"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_contours", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                             "tools", "fuzz_contours.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from rmcv_amd import Context
    rng = np.random.default_rng(2024)
    c = Context(device=0, max_frames=1, max_width=512, max_height=512, max_contours=8192, max_points=1 << 17)
    fast = 0
    for t in range(300):
        h, w = int(rng.integers(8, 300)), int(rng.integers(8, 400))
        canvas = fz.random_scene(rng, h, w)
        _check_contours(c, canvas, oracle)
        fast += int(c.counts()["status"][0]) == 0
    assert fast > 200
    c.close()


@pytest.mark.parametrize("loose", [0, 1])
def test_full_path_fuzz_random_scenes(oracle, loose):
    """320 random scenes (tools/fuzz_path.py) through the whole path, default gates and every gate wide open (each contour with
    >= 6 points fitted -- discs, rings, lines, blobs of noise -- and nearly every pair of blobs built into an armour)"""
    import importlib.util
    import os
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    sys.path.insert(0, tools)
    spec = importlib.util.spec_from_file_location("fuzz_path", os.path.join(tools, "fuzz_path.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.main(10, 100 + loose, loose) == 0
