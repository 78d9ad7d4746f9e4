"""dev tool: is a gigabyte of frames 'fast' or 'slow' by itself?  16 frame sets in one pool; two contexts read ONE set over and over (two
streams; warm: the Infinity Cache helps every set alike); then groups of four sets as tools/placement_probe.py measures them."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, OPT_PIXEL_GROUPS, STAGE_BINARY, Context, default_params, synth  # noqa: E402

n, W, H, NS = 256, 1280, 1024, 16
torch.cuda.init()
host = [synth.batch(k * 1000003, n, W, H, CAMP_BLUE, 0, threads=16) for k in range(4)]
pool = torch.empty((NS, n, H, W, 3), dtype=torch.uint8, device="cuda")
for s in range(NS):
    pool[s].copy_(torch.from_numpy(host[s % 4]))
ctxs = [Context(device=0, max_frames=n, max_width=W, max_height=H) for _ in range(4)]
for c in ctxs:
    c.set_option(OPT_PIXEL_GROUPS, 2)
p = default_params()
streams = [torch.cuda.Stream() for _ in range(2)]


def run(sets, K=80):
    for k, c in enumerate(ctxs):
        fr = pool[sets[k % len(sets)]]
        c.bind_device_frames(fr.data_ptr(), n, H, W, keepalive=fr)
    out = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            ctxs[i % 4].run(p, STAGE_BINARY, streams[i % 2].cuda_stream)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / K * 1e3)
    return float(np.median(out))


print("pool at %x" % pool.data_ptr())
single = [run([s]) for s in range(NS)]
print("one set, read over and over:", " ".join("%.4f" % x for x in single))
for g in range(NS // 4):
    sets = list(range(4 * g, 4 * g + 4))
    print("sets %s in turn: %.4f   (mean of their single figures %.4f)" % (sets, run(sets), float(np.mean([single[s] for s in sets]))))
order = np.argsort(single)
print("the four fastest sets %s in turn: %.4f" % (list(order[:4]), run(list(order[:4]))))
print("the four slowest sets %s in turn: %.4f" % (list(order[-4:]), run(list(order[-4:]))))
