"""dev tool: phase timings of a -DRMCV_PROFILE build of the sparse kernel (RMCV_LIB_PATH=.../var_prof.so), alone and beside 1 GiB copies"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmcv_amd import CAMP_BLUE, OPT_SPARSE_WAVES, STAGE_ALL, STAGE_BINARY, Context, default_params, synth
torch.cuda.init()
n = 128
frames = torch.from_numpy(synth.batch(0, n, 1280, 1024, CAMP_BLUE, 0, threads=16)).cuda()
sA = torch.cuda.Stream(priority=-1); sB = torch.cuda.Stream(priority=0)
big_a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); big_b = torch.empty_like(big_a)
p = default_params()
c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
c.set_option(OPT_SPARSE_WAVES, int(sys.argv[1]) if len(sys.argv) > 1 else 4)
c.bind_device_frames(frames.data_ptr(), n, 1024, 1280, keepalive=frames)
c.run(p, STAGE_ALL, sA.cuda_stream); torch.cuda.synchronize()
for name in ("alone", "beside copies"):
    print("==", name, flush=True)
    if name != "alone":
        with torch.cuda.stream(sB):
            for _ in range(6): big_b.copy_(big_a, non_blocking=True)
    c.run(p, STAGE_ALL & ~STAGE_BINARY, sA.cuda_stream)
    torch.cuda.synchronize()
