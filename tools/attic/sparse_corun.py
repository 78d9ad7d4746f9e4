"""dev tool: what slows the fused sparse kernel down beside other work?  The sparse kernel (4 wavefronts per frame, planes of a finished
batch) runs 40 times on a high-priority stream while a second stream keeps the GPU busy with (a) nothing, (b) large device-to-device
copies (HBM traffic, hardly any instructions), (c) transcendental element-wise work on an L2-resident tensor (instructions, hardly any
HBM traffic), (d) the pixel kernel of another context.  HIP events around the 40 sparse launches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmcv_amd import CAMP_BLUE, OPT_SPARSE_WAVES, OPT_PIXEL_GROUPS, STAGE_ALL, STAGE_BINARY, Context, default_params, synth
torch.cuda.init()
n = 256
frames = torch.from_numpy(synth.batch(0, n, 1280, 1024, CAMP_BLUE, 0, threads=16)).cuda()
frames2 = torch.from_numpy(synth.batch(1000, n, 1280, 1024, CAMP_BLUE, 0, threads=16)).cuda()
p = default_params()
sA = torch.cuda.Stream(priority=-1)
sB = torch.cuda.Stream(priority=0)
waves = int(sys.argv[1]) if len(sys.argv) > 1 else 4
c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
c.set_option(OPT_SPARSE_WAVES, waves)
c.bind_device_frames(frames.data_ptr(), n, 1024, 1280, keepalive=frames)
c.run(p, STAGE_ALL, sA.cuda_stream)
c2 = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
c2.set_option(OPT_PIXEL_GROUPS, int(sys.argv[2]) if len(sys.argv) > 2 else 2)
c2.bind_device_frames(frames2.data_ptr(), n, 1024, 1280, keepalive=frames2)
c2.run(p, STAGE_BINARY, sB.cuda_stream)
torch.cuda.synchronize()
big_a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); big_b = torch.empty_like(big_a)
small = torch.rand(1 << 20, device="cuda")  # 4 MB: L2-resident
def hog_none(k): pass
def hog_copy(k):
    for _ in range(k): big_b.copy_(big_a, non_blocking=True)
def hog_alu(k):
    for _ in range(k * 40): small.sin_()
def hog_pixel(k):
    for _ in range(k * 2): c2.run(p, STAGE_BINARY, sB.cuda_stream)
R = 40
for name, hog, k in (("alone", hog_none, 0), ("beside 1 GiB copies (HBM)", hog_copy, 40), ("beside sin() on 4 MB (ALU)", hog_alu, 40), ("beside k_binary", hog_pixel, 40)):
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sB):
            h0.record(sB)
            hog(k)
            h1.record(sB)
        with torch.cuda.stream(sA):
            e0.record(sA)
            for _ in range(R): c.run(p, STAGE_ALL & ~STAGE_BINARY, sA.cuda_stream)
            e1.record(sA)
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1) / R, h0.elapsed_time(h1)))
    ts.sort()
    print("waves %d sparse kernel %-28s median %.4f ms per launch (the other stream was busy for %.1f ms of the %.1f)" % (waves, name, ts[2][0], ts[2][1], ts[2][0] * R))
