"""dev tool: where one frame's time goes -- kernel durations of a 1-frame batch (HIP events) next to the per-call wall times of the
per-frame chain"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, STAGE_ALL, STAGE_BINARY, STAGE_CONTOURS, Context, default_params, synth  # noqa: E402

torch.cuda.init()
W, H = 1280, 1024
img = synth.frame(3, W, H, CAMP_BLUE, 0)
c = Context(device=0, max_frames=1, max_width=W, max_height=H)
c.upload(img[None])
p = default_params()
for _ in range(10):
    c.run_timed(p, STAGE_ALL)
ms = np.median([c.run_timed(p, STAGE_ALL) for _ in range(50)], axis=0)
print("1-frame batch, kernel durations (ms): binary %.4f contours %.4f blobs %.4f armours %.4f total %.4f" % tuple(ms))
ms = np.median([c.run_timed(p, STAGE_BINARY | STAGE_CONTOURS) for _ in range(50)], axis=0)
print("binary+contours only: binary %.4f contours %.4f" % (ms[0], ms[1]))
for name, fn in (("extract_color", lambda: c.extract_color_csr(img)),):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(50):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%s via Python: median %.4f ms" % (name, np.median(ts)))
t = torch.from_numpy(img)
d = torch.empty_like(t, device="cuda")
b = torch.empty((H, W), dtype=torch.uint8, device="cuda")
hb = torch.empty((H, W), dtype=torch.uint8)
for nm, fn in (("H2D 3.9 MB pageable", lambda: d.copy_(t)), ("D2H 1.3 MB pageable", lambda: hb.copy_(b))):
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%s: median %.4f ms" % (nm, np.median(ts)))
