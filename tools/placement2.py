"""dev tool: where a frame gigabyte lives decides k_binary's level (HISTORY 6g).  ONE context (its outputs stay where they are), N frame buffers
allocated one after the other, each bound in turn, one synchronised launch at a time (cold: the rotation is N GB): the level BY BUFFER.
PAD_GB allocates that much first; ORDER=rev times them in reverse order; WS selects the kernel (RMCV_OPT_PIXEL_SHAPE)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, OPT_PIXEL_GROUPS, STAGE_BINARY, Context, default_params, synth  # noqa: E402

n, W, H = 256, 1280, 1024
N = int(os.environ.get("NBUF", 16))
torch.cuda.init()
dev = torch.device("cuda", 0)
pad = int(os.environ.get("PAD_GB", 0))
keep = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(pad)]
ctx_first = os.environ.get("CTX_FIRST", "0") == "1"
c = Context(device=0, max_frames=n, max_width=W, max_height=H) if ctx_first else None
host = torch.from_numpy(synth.batch(7, n, W, H, CAMP_BLUE, 0, threads=16))
bufs = []
for k in range(N):
    b = torch.empty_like(host, device=dev)
    b.copy_(host)
    bufs.append(b)
if c is None:
    c = Context(device=0, max_frames=n, max_width=W, max_height=H)
c.set_option(OPT_PIXEL_GROUPS, 3)
c.set_option(14, 1 if int(os.environ.get("WS", 0)) else 0)  # RMCV_OPT_PIXEL_SHAPE
p = default_params()
s = torch.cuda.Stream()
d = np.zeros((N, 8))
for rep in range(10):
    for k in range(N):
        c.bind_device_frames(bufs[k].data_ptr(), n, H, W, keepalive=bufs[k])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            a.record(s)
            c.run(p, STAGE_BINARY, s.cuda_stream)
            b.record(s)
        s.synchronize()
        if rep >= 2:
            d[k, rep - 2] = a.elapsed_time(b)
print("pad %d GB, ctx %s, kernel %s" % (pad, "first" if ctx_first else "last", os.environ.get("WS", 0)))
for k in range(N):
    print("buf %2d  %#x  %.4f ms" % (k, bufs[k].data_ptr(), np.median(d[k])))
