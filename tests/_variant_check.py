"""helper of test_gpu_variants.py: run in a fresh process (the library reads its dev knobs once per process);
compares extract_color against the oracle on a few images and exits non-zero on a mismatch"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
from rmcv_amd import CAMP_BLUE, MORPH_CLOSE, STAGE_ALL, Context, default_params, synth  # noqa: E402

O.set_math_mode(0)
ctx = Context(device=0, max_frames=4)
rng = np.random.default_rng(3)
imgs = []
for (h, w) in [(64, 64), (120, 200), (256, 256)]:
    c = ((rng.random((h, w)) < 0.15) * 255).astype(np.uint8)
    c[h // 4:h // 2, w // 4:w // 2] = 255
    c[h // 4 + 3:h // 2 - 3, w // 4 + 3:w // 2 - 3] = 0
    img = np.zeros((h, w, 3), np.uint8)
    img[..., 0] = c
    imgs.append(img)
for img in imgs:
    pts, offs, binary = ctx.extract_color_csr(img, CAMP_BLUE, 80, MORPH_CLOSE)
    rb = O.extract_binary(img, CAMP_BLUE, 80, MORPH_CLOSE)
    rp, ro = O.find_contours(rb)
    assert np.array_equal(binary, rb) and np.array_equal(offs, ro) and np.array_equal(pts, rp), img.shape
frames = synth.batch(300, 4, 1280, 1024, CAMP_BLUE, 1)
arm, offs = ctx.detect_batch(frames, default_params())
st = ctx.counts()["status"]
for f in range(4):
    ref = O.detect_frame(frames[f])
    assert arm[offs[f]:offs[f + 1]].tobytes() == ref["armours"].tobytes(), f
print("variant ok; slow-path frames:", int(np.count_nonzero(st & 16)), "mid-tier frames:", int(np.count_nonzero(st & 64)))
