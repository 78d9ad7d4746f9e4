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
    assert not cnt["status"].any() or set(np.unique(cnt["status"])) <= {0, 16}
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
