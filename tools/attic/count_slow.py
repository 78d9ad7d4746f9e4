# dev tool: how many frames of the synthetic batches take the literal contour scanner (status bit 16)?
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rmcv_amd import CAMP_BLUE, STAGE_ALL, Context, default_params, synth
n = 256
c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
for variant in (0, 1):
    frames = synth.batch(variant * 5000, n, 1280, 1024, CAMP_BLUE, variant, threads=16)
    c.upload(frames)
    c.run(default_params(), STAGE_ALL)
    c.sync()
    st = c.counts()["status"]
    print("variant", variant, "slow-path frames", int(np.count_nonzero(st & 16)), "of", n, "other bits", int(np.count_nonzero(st & ~16)))
