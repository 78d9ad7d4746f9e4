"""worker of tests/test_gpu_export.py: the per-frame chain (three C-ABI calls, run-ahead on) over a few frames; prints one sha256 over
everything the calls returned.  RMCV_EXPORT (read once per process by the library) selects how the results travel."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rmcv_amd import CAMP_BLUE, MORPH_CLOSE, Context, synth

h = hashlib.sha256()
c = Context(device=0, max_frames=1, max_width=1280, max_height=1024)
for i in range(12):
    img = synth.frame(400 + i, 1280, 1024, CAMP_BLUE, 1 if i % 3 == 2 else (14 if i == 7 else 0))   # plain, stress and one dense frame
    pts, offs, binary = c.extract_color_csr(img, CAMP_BLUE, 80, MORPH_CLOSE)
    pos, src, neg = c.filter_lightblobs(pts, offs, 70.0, (1.5, 80.0), (10.0, 99999.0), CAMP_BLUE)
    arm = c.filter_armours(pos, 12.0, 22.0, 0.4, CAMP_BLUE)
    for a in (binary, pts, offs, pos, src, neg, arm):
        h.update(np.asarray(len(a), np.int64).tobytes())
        h.update(np.ascontiguousarray(a).tobytes())
print("chain", h.hexdigest())
