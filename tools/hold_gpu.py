# dev tool (tools/chain_ab.sh): a process that holds a pipeline's worth of HIP queues on GPU 0, runs a few batches, then idles for argv[1] seconds
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmcv_amd import CAMP_BLUE, STAGE_ALL, Pipeline, default_params, synth
n, w, h = 64, 1280, 1024
pl = Pipeline(device=0, max_frames=n, max_width=w, max_height=h)
d = torch.from_numpy(synth.batch(1, n, w, h, CAMP_BLUE, 0, threads=8)).to("cuda:0")
for _ in range(24):
    pl.submit(d.data_ptr(), n, h, w, default_params(), STAGE_ALL)
pl.drain()
print("[hold_gpu] idle with", pl.info.depth, "contexts", flush=True)
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 30)
pl.close()
