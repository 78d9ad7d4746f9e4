"""dev tool: the wave-specialised pixel kernel (k_binary_ws.inc, dev option 1002) next to k_binary, alone and COLD (4 contexts on 4 frame
sets in turn): same results (byte image of every frame, contours and armours of the whole batch), then the times, one stream and two."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, OPT_PIXEL_GROUPS, STAGE_ALL, STAGE_BINARY, Context, default_params, synth  # noqa: E402

n = int(os.environ.get("N", 256))
W, H = int(os.environ.get("W", 1280)), int(os.environ.get("H", 1024))
dense = int(os.environ.get("DENSE", 0))
torch.cuda.init()
dev = torch.device("cuda", 0)
sets = [torch.from_numpy(synth.batch(k * 1000003, n, W, H, CAMP_BLUE, dense, threads=16)).to(dev) for k in range(4)]
ctxs = []
for k in range(4):
    c = Context(device=0, max_frames=n, max_width=W, max_height=H)
    c.bind_device_frames(sets[k].data_ptr(), n, H, W, keepalive=sets[k])
    ctxs.append(c)
p = default_params()

if os.environ.get("CHECK", "1") != "0":
    for morph in (2, 1, 0):
        p.morph = morph
        res = []
        for ws in (0, int(os.environ.get("WS", 1))):
            c = ctxs[0]
            c.set_option(1002, ws)
            c.run(p, STAGE_ALL)
            c.sync()
            b = [c.binary(f).copy() for f in range(0, n, max(1, n // 16))] + [c.binary(n - 1).copy()]
            cont = [c.contours(f) for f in (0, n // 2, n - 1)]
            arm = c.armours()
            res.append((b, cont, arm))
        same_b = all(np.array_equal(x, y) for x, y in zip(res[0][0], res[1][0]))
        same_c = all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(res[0][1], res[1][1]))
        same_a = res[0][2][0].tobytes() == res[1][2][0].tobytes() and np.array_equal(res[0][2][1], res[1][2][1])
        print("morph %d: byte images equal %s, contours equal %s, armours equal %s (%d armours)" % (morph, same_b, same_c, same_a, len(res[0][2][0])), flush=True)
    p.morph = 2

ss = [torch.cuda.Stream(), torch.cuda.Stream()]
shapes = {1: "8+8 U2", 2: "8+8 U3", 3: "8+8 U4", 4: "12+4 U2", 5: "8+4 U2", 6: "4+4 U4 x2", 7: "6+2 U3 x2", 8: "10+4 U2", 9: "ring 12+4 R2", 10: "ring 12+4 R3", 11: "ring 8+8 R3", 12: "ring 8+4 R3", 13: "ring 10+4 R3",
          14: "ring 8+4 R4", 15: "ring 8+4 R2", 16: "ring 8+8 R2", 17: "ring 12+4 R4", 18: "ring 8+8 R4", 19: "ring 12+4 R3 cacheable", 20: "ring 8+8 R3 cacheable", 21: "ring 8+8 R5"}
if os.environ.get("SHAPES"):
    shapes = {int(v): shapes[int(v)] for v in os.environ["SHAPES"].split(",")}
runs = [("k_binary 2/CU", 0, 2)] + [("ws " + shapes[v], v, 2) for v in sorted(shapes)] + [("k_binary 2/CU", 0, 2), ("ws " + shapes[1], 1, 2)]
for label, ws, groups in runs:
    for c in ctxs:
        c.set_option(OPT_PIXEL_GROUPS, groups)
        c.set_option(1002, ws)
    out = []
    for nstreams in (1, 2):
        for i in range(8):
            ctxs[i % 4].run(p, STAGE_BINARY, ss[i % nstreams].cuda_stream)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for i in range(40):
                ctxs[i % 4].run(p, STAGE_BINARY, ss[i % nstreams].cuda_stream)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 40 * 1e3)
        ts.sort()
        out.append(ts[2])
    print("%-16s one stream %.4f ms (%.0f GB/s)   two streams %.4f ms (%.0f GB/s)" %
          (label, out[0], n * W * H * 4 / out[0] / 1e6, out[1], n * W * H * 4 / out[1] / 1e6), flush=True)
