"""dev tool: one pixel kernel at a time, COLD (8 contexts on 8 frame sets in turn), HIP events around every launch: the duration BY CONTEXT,
k_binary next to k_binary_ws (RMCV_OPT_PIXEL_SHAPE).  Do some contexts' buffers sit badly?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, OPT_PIXEL_GROUPS, STAGE_BINARY, STAGE_NO_IMAGE, Context, default_params, synth  # noqa: E402

n, W, H = 256, 1280, 1024
NC = int(os.environ.get("NCTX", 8))
torch.cuda.init()
dev = torch.device("cuda", 0)
sets = [torch.from_numpy(synth.batch(k * 1000003, n, W, H, CAMP_BLUE, 0, threads=16)).to(dev) for k in range(NC)]
ctxs = []
for k in range(NC):
    c = Context(device=0, max_frames=n, max_width=W, max_height=H)
    c.bind_device_frames(sets[k].data_ptr(), n, H, W, keepalive=sets[k])
    ctxs.append(c)
p = default_params()
ST = STAGE_BINARY | (STAGE_NO_IMAGE if os.environ.get("NO_IMAGE") else 0)
SAME_SET = os.environ.get("SAME_SET")
if SAME_SET:
    for c in ctxs:
        c.bind_device_frames(sets[0].data_ptr(), n, H, W, keepalive=sets[0])
s = torch.cuda.Stream()
for label, ws in (("k_binary", 0), ("ws", int(os.environ.get("WS", 11))), ("k_binary", 0), ("ws", int(os.environ.get("WS", 11)))):
    for c in ctxs:
        c.set_option(14, 1 if ws else 0)  # RMCV_OPT_PIXEL_SHAPE
        c.set_option(OPT_PIXEL_GROUPS, 3)
    d = np.zeros((NC, 12))
    for rep in range(14):
        for k, c in enumerate(ctxs):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s):
                a.record(s)
                c.run(p, ST, s.cuda_stream)
                b.record(s)
            s.synchronize()
            if rep >= 2:
                d[k, rep - 2] = a.elapsed_time(b)
    print("%-9s by context (median ms): %s   all: %.4f" % (label, " ".join("%.4f" % np.median(d[k]) for k in range(NC)), np.median(d)), flush=True)
views = [c.device_views() for c in ctxs]
for k, c in enumerate(ctxs):
    print("ctx %d frames %#x  %s" % (k, sets[k].data_ptr(), {kk: hex(v) if isinstance(v, int) else v for kk, v in list(views[k].items())[:8]} if isinstance(views[k], dict) else views[k]))
