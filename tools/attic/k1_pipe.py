"""dev tool: k_binary ONLY, pipelined the way bench.py's loop launches it but with nothing else on the machine and no events between
the launches: C contexts with their own frames, launches alternating over S streams, G workgroups per CU.  What is the best the
two-stream overlap can do, against one stream back to back?   python tools/k1_pipe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, OPT_PIXEL_GROUPS, STAGE_BINARY, Context, default_params, synth  # noqa: E402

n, W, H = 256, 1280, 1024
torch.cuda.init()
NC = 4
frames = [torch.from_numpy(synth.batch(k * 1000003, n, W, H, CAMP_BLUE, 0, threads=16)).cuda() for k in range(NC)]
ctxs = []
for k in range(NC):
    c = Context(device=0, max_frames=n, max_width=W, max_height=H)
    c.bind_device_frames(frames[k].data_ptr(), n, H, W, keepalive=frames[k])
    ctxs.append(c)
p = default_params()
# U = how many of the contexts (each with its own 1 GB of frames and its own output buffers) the launches rotate over: 1 = every launch
# reads the same gigabyte again (part of it still in the 256 MB Infinity Cache), 4 = cold
CASES = ((1, 1, 3), (2, 1, 3), (4, 1, 3), (1, 1, 2), (4, 1, 2), (1, 2, 2), (2, 2, 2), (4, 2, 2), (4, 2, 3), (4, 3, 2), (1, 1, 3))
if len(sys.argv) > 1 and sys.argv[1] == "groups":       # cold, one stream: workgroups per CU
    CASES = tuple((4, 1, g) for g in (1, 2, 3, 4, 5, 6, 8)) + tuple((4, 2, g) for g in (1, 2, 3, 4))
if len(sys.argv) > 1 and sys.argv[1] == "short":
    CASES = ((4, 1, 3), (4, 1, 2), (4, 2, 2), (1, 1, 3))
for U, S, G in CASES:
    streams = [torch.cuda.Stream() for _ in range(S)]
    for c in ctxs:
        c.set_option(OPT_PIXEL_GROUPS, G)
    res = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 200
        for i in range(K):
            ctxs[i % U].run(p, STAGE_BINARY, streams[i % S].cuda_stream)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / K * 1e3)
    res.sort()
    print("frame sets %d  streams %d  groups %d: median %.4f ms per launch  min %.4f" % (U, S, G, res[len(res) // 2], res[0]), flush=True)
