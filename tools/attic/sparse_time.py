"""dev tool: the fused sparse kernel alone (both settings) and the lone-batch latency, HIP events"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmcv_amd import CAMP_BLUE, OPT_SPARSE_WAVES, OPT_PIXEL_GROUPS, STAGE_ALL, STAGE_BINARY, Context, default_params, synth
torch.cuda.init()
n = 256
frames = torch.from_numpy(synth.batch(0, n, 1280, 1024, CAMP_BLUE, 0, threads=16)).cuda()
p = default_params()
s = torch.cuda.Stream()
for waves in (8, 4):
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    c.set_option(OPT_SPARSE_WAVES, waves)
    c.bind_device_frames(frames.data_ptr(), n, 1024, 1280, keepalive=frames)
    c.run(p, STAGE_ALL, s.cuda_stream)
    torch.cuda.synchronize()
    for stages, name in ((STAGE_ALL & ~STAGE_BINARY, "fused sparse kernel"), (STAGE_ALL, "whole batch (binary + sparse)")):
        ts = []
        for rep in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s):
                e0.record(s)
                for _ in range(10):
                    c.run(p, stages, s.cuda_stream)
                e1.record(s)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        print("waves %d %-32s median %.4f ms min %.4f" % (waves, name, np.median(ts), min(ts)))
    c.close()
