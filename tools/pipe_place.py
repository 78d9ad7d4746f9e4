# dev tool: does the step time depend on WHICH hardware queues the pipeline's streams get?  `offset` dummy streams are created and
# used first (each takes a hardware queue), then the pipeline; one process per offset (the mapping is per process).
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmcv_amd import CAMP_BLUE, STAGE_ALL, Pipeline, default_params, synth
offset = int(sys.argv[1])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, W, H = 256, 1280, 1024
sets = [torch.from_numpy(synth.batch(k * 1000003, n, W, H, CAMP_BLUE, 0, threads=16)).to(dev) for k in range(8)]
dummies = [torch.cuda.Stream(device=dev) for _ in range(offset)]
x = torch.zeros(1024, device=dev)
for s in dummies:
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
p = default_params()
i = [0]
def region(pl, k=200):
    pl.drain()
    t0 = time.perf_counter()
    for _ in range(k): pl.submit(sets[i[0] % 8].data_ptr(), n, H, W, p, STAGE_ALL); i[0] += 1
    pl.drain()
    return (time.perf_counter() - t0) / k * 1e3
pl = Pipeline(device=0, depth=8, armour_cap=n * 8, max_frames=n, max_width=W, max_height=H)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.6: region(pl, 100)
r = sorted(region(pl) for _ in range(5))
print("offset %d: median %.4f min %.4f max %.4f" % (offset, r[2], r[0], r[4]), flush=True)
