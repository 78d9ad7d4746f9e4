"""dev tool: the wave-specialised pixel kernel (k_binary_ws.inc, RMCV_OPT_PIXEL_SHAPE 1) next to k_binary, alone and COLD (NCTX contexts
on NSETS frame sets in turn): same results (byte image of sampled frames, contours and armours of the whole batch, every morph), then
the times, one stream and two.  NCTX=8 shows what the number of bit planes in rotation does (profiles/r04f_k_binary_ws.txt)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, OPT_PIXEL_GROUPS, OPT_PIXEL_SHAPE, STAGE_ALL, STAGE_BINARY, STAGE_NO_IMAGE, Context, default_params, synth  # noqa: E402

STB = STAGE_BINARY | (STAGE_NO_IMAGE if os.environ.get("NO_IMAGE") else 0)
n = int(os.environ.get("N", 256))
W, H = int(os.environ.get("W", 1280)), int(os.environ.get("H", 1024))
NC = int(os.environ.get("NCTX", 4))
REG = int(os.environ.get("REG", 40))
torch.cuda.init()
dev = torch.device("cuda", 0)
sets = [torch.from_numpy(synth.batch(k * 1000003, n, W, H, CAMP_BLUE, 0, threads=16)).to(dev) for k in range(int(os.environ.get("NSETS", NC)))]
ctxs = []
for k in range(NC):
    c = Context(device=0, max_frames=n, max_width=W, max_height=H)
    c.bind_device_frames(sets[k % len(sets)].data_ptr(), n, H, W, keepalive=sets[k % len(sets)])
    ctxs.append(c)
p = default_params()

if os.environ.get("CHECK", "1") != "0":
    for morph in (2, 1, 0):
        p.morph = morph
        res = []
        for ws in (0, 1):
            c = ctxs[0]
            c.set_option(OPT_PIXEL_SHAPE, ws)
            c.run(p, STAGE_ALL)
            c.sync()
            b = [c.binary(f).copy() for f in range(0, n, max(1, n // 16))] + [c.binary(n - 1).copy()]
            cont = [c.contours(f) for f in (0, n // 2, n - 1)]
            res.append((b, cont, c.armours()))
        same_b = all(np.array_equal(x, y) for x, y in zip(res[0][0], res[1][0]))
        same_c = all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(res[0][1], res[1][1]))
        same_a = res[0][2][0].tobytes() == res[1][2][0].tobytes() and np.array_equal(res[0][2][1], res[1][2][1])
        print("morph %d: byte images equal %s, contours equal %s, armours equal %s (%d armours)" % (morph, same_b, same_c, same_a, len(res[0][2][0])), flush=True)
    p.morph = 2

ss = [torch.cuda.Stream(), torch.cuda.Stream()]
for label, ws, groups in (("k_binary 2/CU", 0, 2), ("k_binary 3/CU", 0, 3), ("k_binary_ws", 1, 2), ("k_binary 2/CU", 0, 2), ("k_binary_ws", 1, 2)):
    for c in ctxs:
        c.set_option(OPT_PIXEL_GROUPS, groups)
        c.set_option(OPT_PIXEL_SHAPE, ws)
    out = []
    for nstreams in (1, 2):
        for i in range(2 * NC):
            ctxs[i % NC].run(p, STB, ss[i % nstreams].cuda_stream)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for i in range(REG):
                ctxs[i % NC].run(p, STB, ss[i % nstreams].cuda_stream)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / REG * 1e3)
        ts.sort()
        out.append(ts[2])
    print("%-16s %d contexts: one stream %.4f ms (%.0f GB/s)   two streams %.4f ms (%.0f GB/s)" %
          (label, NC, out[0], n * W * H * 4 / out[0] / 1e6, out[1], n * W * H * 4 / out[1] / 1e6), flush=True)
