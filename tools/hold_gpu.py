# dev tool (tools/chain_ab.sh): a process that holds what bench.py holds when it starts its C-host child -- several full-size pipelines
# (argv[2], default 3), gigabytes of resident frames --, runs a few batches on each, then idles for argv[1] seconds
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmcv_amd import CAMP_BLUE, STAGE_ALL, Pipeline, default_params, synth
n, w, h = 256, 1280, 1024
npl = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sets = [torch.from_numpy(synth.batch(1000 * k, n, w, h, CAMP_BLUE, 0, threads=16)).to("cuda:0") for k in range(8)]
pls = [Pipeline(device=0, max_frames=n, max_width=w, max_height=h) for _ in range(npl)]
for pl in pls:
    for i in range(24):
        pl.submit(sets[i % 8].data_ptr(), n, h, w, default_params(), STAGE_ALL)
    pl.drain()
print("[hold_gpu] idle with %d pipelines x %d contexts, %.1f GB of frames" % (npl, pls[0].info.depth, 8 * n * w * h * 3 / 1e9), flush=True)
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 30)
for pl in pls:
    pl.close()
