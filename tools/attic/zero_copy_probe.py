"""dev tool: k_binary reading ONE frame straight out of pinned host memory (no upload) against upload + k_binary from HBM."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, STAGE_BINARY, Context, default_params, synth  # noqa: E402

W, H = 1280, 1024
torch.cuda.init()
host = torch.from_numpy(synth.batch(0, 4, W, H, CAMP_BLUE, 0, threads=4)).pin_memory()
dev = torch.empty_like(host, device="cuda")
c = Context(device=0, max_frames=1, max_width=W, max_height=H)
p = default_params()
s = torch.cuda.Stream()


def t_upload_then_kernel(k):
    with torch.cuda.stream(s):
        dev[k].copy_(host[k], non_blocking=True)
        c.bind_device_frames(dev[k].data_ptr(), 1, H, W, keepalive=dev)
        c.run(p, STAGE_BINARY, s.cuda_stream)


def t_zero_copy(k):
    c.bind_device_frames(host[k].data_ptr(), 1, H, W, keepalive=host)
    c.run(p, STAGE_BINARY, s.cuda_stream)


for name, fn in (("upload + k_binary", t_upload_then_kernel), ("k_binary reading pinned host memory", t_zero_copy), ("upload + k_binary", t_upload_then_kernel)):
    ts = []
    for i in range(60):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(i % 4)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.sort(ts[10:])
    print("%-40s median %.4f ms  min %.4f" % (name, ts[len(ts) // 2], ts[0]))
    ref = c.binary(0).astype(np.int64).sum()
    print("   checksum", int(ref))
