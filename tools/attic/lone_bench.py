"""dev tool: latency of ONE batch (256 x 1280x1024, full path + compaction) on one stream, by pixel workgroups per CU, sparse waves per
frame and frame-level hand-over.   python tools/lone_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, OPT_HANDOVER, OPT_PIXEL_GROUPS, OPT_SPARSE_WAVES, STAGE_ALL, Context, default_params, synth  # noqa: E402
from rmcv_amd import dist as rdist  # noqa: E402

n, W, H = 256, 1280, 1024
torch.cuda.init()
frames = torch.from_numpy(synth.batch(0, n, W, H, CAMP_BLUE, int(sys.argv[1]) if len(sys.argv) > 1 else 0, threads=16)).cuda()
c = Context(device=0, max_frames=n, max_width=W, max_height=H, max_contours=4096)
c.bind_device_frames(frames.data_ptr(), n, H, W, keepalive=frames)
p = default_params()
s = torch.cuda.Stream()
cap = n * 8
head, _ = rdist.record_layout(n, cap)
rec = rdist.new_record(n, cap, frames.device)
one = len(sys.argv) > 2
for ho in ((0, 1) if one else (1, 0)):
    for groups in ((3,) if one else (2, 3, 4)):
        for waves in ((4,) if one else (4, 8)):
            c.set_option(OPT_HANDOVER, ho)
            c.set_option(OPT_PIXEL_GROUPS, groups)
            c.set_option(OPT_SPARSE_WAVES, waves)
            ts = []
            for rep in range(8 if one else 25):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(s):
                    e0.record(s)
                    c.run(p, STAGE_ALL, s.cuda_stream)
                    c.compact_armours_into(rec.data_ptr() + head, cap, rec.data_ptr(), s.cuda_stream)
                    e1.record(s)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts = sorted(ts[2:] if one else ts[5:])
            print("hand-over %d  pixel groups %d  sparse waves %d: median %.4f ms  min %.4f" % (ho, groups, waves, ts[len(ts) // 2], ts[0]), flush=True)
