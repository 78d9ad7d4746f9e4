"""dev tool: k_binary alone, back to back on 256 resident frames (HIP events on the launch stream); for same-box A/B runs of
library builds (RMCV_LIB_PATH) and environment knobs:  python tools/k1_bench.py [groups] [morph] [w h]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, OPT_PIXEL_GROUPS, STAGE_BINARY, STAGE_NO_IMAGE, Context, default_params, synth  # noqa: E402

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 4
morph = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1280, 1024)
n = 256
torch.cuda.init()
frames = torch.from_numpy(synth.batch(0, n, W, H, CAMP_BLUE, 0, threads=16)).cuda()
c = Context(device=0, max_frames=n, max_width=W, max_height=H)
c.set_option(OPT_PIXEL_GROUPS, groups)
c.bind_device_frames(frames.data_ptr(), n, H, W, keepalive=frames)
p = default_params(morph=morph)
s = torch.cuda.Stream()
for stages, name in ((STAGE_BINARY, "image"), (STAGE_BINARY | STAGE_NO_IMAGE, "no-image")):
    for _ in range(30):
        c.run(p, stages, s.cuda_stream)
    torch.cuda.synchronize()
    ts = []
    for rep in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record(s)
            for _ in range(20):
                c.run(p, stages, s.cuda_stream)
            e1.record(s)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    ts.sort()
    bytes_ = n * W * H * (4 if name == "image" else 3)
    print("k_binary %-8s groups %d morph %d %dx%d: median %.4f ms  min %.4f  -> %.0f GB/s algorithmic (median)" %
          (name, groups, morph, W, H, ts[len(ts) // 2], ts[0], bytes_ / ts[len(ts) // 2] / 1e6))
chk = int(np.frombuffer(c.binary(3).tobytes(), np.uint8).astype(np.int64).sum())
print("checksum frame 3:", chk)
