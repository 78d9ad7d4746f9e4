"""dev tool: does the pixel kernels' level depend on WHERE a process's buffers landed?  G groups of four contexts (own frames, own buffers)
in ONE process, k_binary only over two streams (tools/k1_pipe.py's 2-stream case), the groups measured in turn several times."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, OPT_PIXEL_GROUPS, STAGE_BINARY, STAGE_NO_IMAGE, Context, default_params, synth  # noqa: E402

n, W, H, G = 256, 1280, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.cuda.init()
host = [synth.batch(k * 1000003, n, W, H, CAMP_BLUE, 0, threads=16) for k in range(4)]
groups = []
ORDER = sys.argv[2] if len(sys.argv) > 2 else "interleaved"   # "first": every group's frames in ONE allocation made before any context exists
pool = None
if ORDER == "first":
    pool = torch.empty((G * 4, n, H, W, 3), dtype=torch.uint8, device="cuda")
    for g in range(G):
        for k in range(4):
            pool[g * 4 + k].copy_(torch.from_numpy(host[k]))
for g in range(G):
    cs = []
    for k in range(4):
        fr = pool[g * 4 + k] if pool is not None else torch.from_numpy(host[k]).cuda()
        c = Context(device=0, max_frames=n, max_width=W, max_height=H)
        c.bind_device_frames(fr.data_ptr(), n, H, W, keepalive=fr)
        c.set_option(OPT_PIXEL_GROUPS, 2)
        cs.append(c)
    groups.append(cs)
p = default_params()
streams = [torch.cuda.Stream() for _ in range(2)]
if len(sys.argv) > 3 and sys.argv[3] == "cross" and pool is not None:   # context group g reads the frames of group g + 1: does the level follow the frames or the contexts?
    for g in range(G):
        for k in range(4):
            fr = pool[((g + 1) % G) * 4 + k]
            groups[g][k].bind_device_frames(fr.data_ptr(), n, H, W, keepalive=fr)
    print("(crossed: context group g reads frame group g + 1)")
for stages, name in ((STAGE_BINARY, "with the byte image"), (STAGE_BINARY | STAGE_NO_IMAGE, "without the byte image (3 B/px)")):
    res = np.zeros((6, G))
    for rep in range(6):
        for g in range(G):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K = 120
            for i in range(K):
                groups[g][i % 4].run(p, stages, streams[i % 2].cuda_stream)
            torch.cuda.synchronize()
            res[rep, g] = (time.perf_counter() - t0) / K * 1e3
    for g in range(G):
        print("%s group %d: median %.4f ms per launch  min %.4f  max %.4f" % (name, g, np.median(res[1:, g]), res[1:, g].min(), res[1:, g].max()))

# is a slow group's gigabyte slow for ANY reader?  a plain reduction over every frame set (torch's own kernel), cold (the sets in turn)
sets = [c._frames_ref for cs in groups for c in cs]
views = [t.reshape(-1).view(torch.int32) for t in sets]
acc = np.zeros(len(views))
for rep in range(6):
    for i, v in enumerate(views):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v.sum()
        torch.cuda.synchronize()
        if rep:
            acc[i] += (time.perf_counter() - t0) * 1e3 / 5
for g in range(G):
    print("group %d frame sets, torch sum of 1 GB: %s ms" % (g, " ".join("%.4f" % x for x in acc[4 * g:4 * g + 4])))
