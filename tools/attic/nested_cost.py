# dev tool: what does the literal contour scanner cost when every frame of a batch has a nested component?
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rmcv_amd import CAMP_BLUE, STAGE_ALL, STAGE_BINARY, Context, default_params, synth
n = 256
frames = synth.batch(0, n, 1280, 1024, CAMP_BLUE, 0, threads=16)
c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
def timeit(tag):
    c.upload(frames)
    p = default_params()
    for _ in range(3):
        c.run(p, STAGE_ALL); c.sync()
    c.run(p, STAGE_BINARY); c.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        c.run(p, STAGE_ALL & ~STAGE_BINARY)
    c.sync()
    dt = (time.perf_counter() - t0) / 10
    st = c.counts()["status"]
    print(tag, "sparse kernel %.3f ms per batch, literal-path frames %d" % (dt * 1e3, int(np.count_nonzero(st & 16))))
timeit("plain stream      ")
for k in (1, 8, 64, 256):
    frames[:] = synth.batch(0, n, 1280, 1024, CAMP_BLUE, 0, threads=16)
    for f in range(0, n, n // k):
        frames[f, 20:80, 20:80] = (255, 60, 0)       # a bright square ...
        frames[f, 30:70, 30:70] = (10, 10, 10)       # ... with a hole ...
        frames[f, 45:55, 45:55] = (255, 60, 0)       # ... and a blob inside the hole
    timeit("%3d nested frames " % k)
