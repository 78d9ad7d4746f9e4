"""dev tool: k_binary only over two streams (four contexts), with the FRAMES placed at different offsets / with a padded frame pitch inside
one big allocation per context: which address relations between the input stream and the context's own buffers cost what?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, OPT_PIXEL_GROUPS, STAGE_BINARY, Context, default_params, synth  # noqa: E402

n, W, H = 256, 1280, 1024
torch.cuda.init()
FB = H * W * 3
host = [torch.from_numpy(synth.batch(k * 1000003, n, W, H, CAMP_BLUE, 0, threads=16)).reshape(n, FB) for k in range(4)]
big = [torch.empty(n * (FB + (1 << 16)) + (64 << 20), dtype=torch.uint8, device="cuda") for _ in range(4)]
ctxs = [Context(device=0, max_frames=n, max_width=W, max_height=H) for _ in range(4)]
for c in ctxs:
    c.set_option(OPT_PIXEL_GROUPS, 2)
p = default_params()
streams = [torch.cuda.Stream() for _ in range(2)]
ref = None


def place(off, pad):
    pitch = FB + pad
    for k in range(4):
        view = big[k][off:off + n * pitch].view(n, pitch)
        view[:, :FB].copy_(host[k].cuda() if False else host[k].to("cuda"))
        ctxs[k].bind_device_frames(big[k].data_ptr() + off, n, H, W, stride=3 * W, frame_pitch=pitch, keepalive=big[k])
    torch.cuda.synchronize()


def measure():
    best = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 120
        for i in range(K):
            ctxs[i % 4].run(p, STAGE_BINARY, streams[i % 2].cuda_stream)
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / K * 1e3)
    return float(np.median(best[1:]))


print("base addresses:", ["%x" % b.data_ptr() for b in big], flush=True)
for off, pad in ((0, 0), (256, 0), (4096, 0), (65536, 0), (1 << 20, 0), (2 << 20, 0), (16 << 20, 0), (0, 256), (0, 4096), (0, 4096 + 256), (0, 65536), (0, 16), (0, 0)):
    place(off, pad)
    ms = measure()
    chk = int(np.frombuffer(ctxs[1].binary(3).tobytes(), np.uint8).astype(np.int64).sum())
    if ref is None:
        ref = chk
    print("offset %9d  pitch pad %6d: %.4f ms per launch   %s" % (off, pad, ms, "ok" if chk == ref else "CHECKSUM DIFFERS"), flush=True)
