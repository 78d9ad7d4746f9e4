"""dev tool: one 256-frame batch at a time through the pipeline (submit; wait) -- the latency a host sees.  RMCV_LAZY_BACK=0: the back half enqueued
by submit itself (4 wavefronts per frame) instead of by the wait (8)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from rmcv_amd import CAMP_BLUE, STAGE_ALL, Pipeline, default_params, synth
n, W, H = 256, 1280, 1024
dev = torch.device("cuda", 0)
sets = [torch.from_numpy(synth.batch(k * 1000003, n, W, H, CAMP_BLUE, 0, threads=16)).to(dev) for k in range(8)]
p = default_params()
pl = Pipeline(device=0, max_frames=n, max_width=W, max_height=H)
ts = []
for rep in range(60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t = pl.submit(sets[rep % 8].data_ptr(), n, H, W, p, STAGE_ALL)
    pl.wait(t)
    ts.append((time.perf_counter() - t0) * 1e3)
ts = sorted(ts[10:])
print("one batch at a time through the pipeline (submit; wait): median %.4f ms min %.4f; latency batches %d, hot %d" % (ts[len(ts)//2], ts[0], pl.get_info().latency_batches, pl.get_info().hot_batches))
