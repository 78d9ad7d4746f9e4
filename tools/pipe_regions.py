# dev tool: ms per step of the pipeline for regions of different lengths (is a long region as fast as a short one?), alternating
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmcv_amd import CAMP_BLUE, STAGE_ALL, Pipeline, default_params, synth
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, W, H = 256, 1280, 1024
host_results = int(sys.argv[1]) if len(sys.argv) > 1 else 1
use_torch_sync = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sets = [torch.from_numpy(synth.batch(k * 1000003, n, W, H, CAMP_BLUE, 0, threads=16)).to(dev) for k in range(8)]
pl = Pipeline(device=0, depth=8, armour_cap=n * 8, host_results=host_results, max_frames=n, max_width=W, max_height=H)
p = default_params()
i = [0]
def step():
    pl.submit(sets[i[0] % 8].data_ptr(), n, H, W, p, STAGE_ALL); i[0] += 1
def region(k):
    pl.drain()
    if use_torch_sync: torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): step()
    te = time.perf_counter() - t0
    pl.drain()
    if use_torch_sync: torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3, te / k * 1e3
for _ in range(400): step()
for rnd in range(2):
    print("host_results", host_results, "torch_sync", use_torch_sync, " ".join("%d: %.4f (enq %.4f)" % ((k,) + region(k)) for k in (20, 100, 500, 20)), flush=True)
