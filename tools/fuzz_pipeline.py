"""dev tool: a long random stream through the pipeline -- plain, noisy and dense batches, geometries and batch sizes changing, partial stage
masks, waits and context getters at random points, the hot contexts switched now and then -- every list checked against the oracle.
    python tools/fuzz_pipeline.py [seconds] [seed]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle_lib as oracle  # noqa: E402
from rmcv_amd import CAMP_BLUE, STAGE_ALL, STAGE_NO_IMAGE, Pipeline, default_params, synth  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
geoms = [(1280, 1024), (1920, 1200), (640, 512), (1280, 720), (1920, 1080), (704, 576)]
NMAX = 64
pl = Pipeline(device=0, max_frames=NMAX, max_width=1920, max_height=1200, max_contours=4096)
p = default_params()
pool = ThreadPoolExecutor(16)
pending = {}          # ticket -> (host frames, device tensor)
checked = batches = 0
t_end = time.time() + secs
geom = geoms[0]
i = 0
last_kind = 0


def check(t):
    global checked
    fr, _ = pending.pop(t)
    arm, offs = pl.collect(t)
    refs = list(pool.map(lambda f: oracle.detect_frame(f)["armours"], fr))
    assert len(offs) == len(fr) + 1, t
    for f, r in enumerate(refs):
        assert arm[offs[f]:offs[f + 1]].tobytes() == r.tobytes(), (t, f)
    checked += 1


while time.time() < t_end:
    if rng.random() < 0.15:
        geom = geoms[int(rng.integers(len(geoms)))]
    w, h = geom
    n = int(rng.choice([NMAX, NMAX, NMAX, 40, 17, 5, 1]))
    kind = int(rng.choice([0, 0, 0, 1, 2, 14, 12, 12, 13]))   # (12, 13: whole batches of dense2 / dense3 frames -- dense mode comes and goes)
    if kind in (12, 13) and rng.random() < 0.7 and i > 0:
        kind = last_kind if last_kind in (12, 13) else kind      # runs of dense batches
    last_kind = kind
    fr = synth.batch(int(rng.integers(1 << 30)), n, w, h, CAMP_BLUE, kind if kind != 14 else 0, threads=16)
    if kind == 14:
        k = max(1, n // int(rng.choice([2, 8, 64])))
        fr[:k] = synth.batch(int(rng.integers(1 << 30)), k, w, h, CAMP_BLUE, 14, threads=16)
    d = torch.from_numpy(fr).to(dev)
    st = STAGE_ALL | (STAGE_NO_IMAGE if rng.random() < 0.2 else 0)
    if rng.random() < 0.05:
        pl.set_hot_contexts(int(rng.choice([0, 3, 4, 5, 7])))
    t = pl.submit(d.data_ptr(), n, h, w, p, st)
    pending[t] = (fr, d)
    batches += 1
    if rng.random() < 0.2:
        pl.wait(t)
        c = pl.context_of(t)
        f = int(rng.integers(n))
        r = oracle.detect_frame(fr[f])
        pts, co = c.contours(f)
        assert np.array_equal(co, r["offs"]) and np.array_equal(pts, r["pts"]), (t, f)
        if not (st & STAGE_NO_IMAGE):
            assert np.array_equal(c.binary(f), r["binary"]), (t, f)
    while len(pending) > int(rng.integers(1, 8)):
        check(min(pending))
    i += 1
pl.drain()
for t in sorted(pending):
    check(t)
info = pl.get_info()
assert info.host_blocking_calls == 0
print("fuzz_pipeline: %d batches, %d lists checked against the oracle, %d ran hot, %d with their dense frames split off, %d in dense mode, 0 blocking calls in submit: all equal" %
      (batches, checked, info.hot_batches, info.dense_split, info.heavy_batches))
pl.close()
