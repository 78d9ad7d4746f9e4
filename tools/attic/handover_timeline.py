"""dev tool: device-side timeline of ONE lone batch with the frame-level hand-over (library built with -DRMCV_PROFILE_HANDOVER:
tools/build_variant_all.sh prof "-DRMCV_PROFILE_HANDOVER"; run with RMCV_LIB_PATH=rmcv_amd/lib/var_prof.so).  Prints, relative to the
pixel kernel's start: when it ended, and for the sparse workgroups the distribution of start / frame-ready / done times."""
import os
import re
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from rmcv_amd import CAMP_BLUE, OPT_HANDOVER, OPT_PIXEL_GROUPS, OPT_SPARSE_WAVES, STAGE_ALL, Context, default_params, synth
    n, W, H = 256, 1280, 1024
    torch.cuda.init()
    frames = torch.from_numpy(synth.batch(0, n, W, H, CAMP_BLUE, 0, threads=16)).cuda()
    c = Context(device=0, max_frames=n, max_width=W, max_height=H)
    c.bind_device_frames(frames.data_ptr(), n, H, W, keepalive=frames)
    c.set_option(OPT_HANDOVER, int(sys.argv[2]))
    c.set_option(OPT_PIXEL_GROUPS, int(sys.argv[3]))
    c.set_option(OPT_SPARSE_WAVES, int(sys.argv[4]))
    s = torch.cuda.Stream()
    for rep in range(4):
        print("== rep", rep, flush=True)
        c.run(default_params(), STAGE_ALL, s.cuda_stream)
        torch.cuda.synchronize()
    sys.exit(0)

import numpy as np
for ho, g, w in [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or ((1, 3, 4), (0, 3, 4), (1, 2, 8)):
    out = subprocess.run([sys.executable, __file__, "child", str(ho), str(g), str(w)], capture_output=True, text=True).stdout
    rep = out.split("== rep 3")[-1]
    kb0 = int(re.search(r"\[kb start\] (\d+)", rep).group(1))
    kb1 = int(re.search(r"\[kb end\] (\d+)", rep).group(1))
    sp = np.array([[int(x) for x in m.groups()] for m in re.finditer(r"\[sp\] (\d+) (\d+) (\d+) (\d+)", rep)], dtype=np.int64)
    ph = np.array([[float(x) for x in m.groups()] for m in re.finditer(r"\[sp\] \d+ \d+ \d+ \d+ \| ([\d.]+) ([\d.]+) ([\d.]+) ([\d.]+) ([\d.]+) \| elig (\d+)", rep)])
    if len(ph):
        tot = ph[:, :5].sum(1)
        order = np.argsort(tot)
        for name, idx in (("median frame", order[len(order) // 2]), ("p90 frame", order[9 * len(order) // 10]), ("slowest frame", order[-1])):
            print("   %-14s tables %.1f  cycles %.1f  verify+rank %.1f  fits %.1f (%d eligible)  compaction+pairing %.1f  = %.1f us" % ((name,) + tuple(ph[idx, :4]) + (int(ph[idx, 5]), ph[idx, 4], tot[idx])))
    us = lambda t: (t - kb0) / 100.0
    print("hand-over %d groups %d waves %d: pixel kernel 0 .. %.1f us; %d sparse workgroups" % (ho, g, w, us(kb1), len(sp)))
    for name, col in (("start", 1), ("frame ready", 2), ("done", 3)):
        v = np.sort(us(sp[:, col]))
        print("   %-12s min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us" % (name, v[0], v[len(v) // 10], v[len(v) // 2], v[9 * len(v) // 10], v[-1]))
    busy = (sp[:, 3] - sp[:, 2]) / 100.0
    print("   work per frame (ready -> done): median %.1f  p90 %.1f  max %.1f us" % (np.median(busy), np.sort(busy)[9 * len(busy) // 10], busy.max()))
