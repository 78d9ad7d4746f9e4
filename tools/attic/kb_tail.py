"""dev tool: when do the workgroups of one k_binary launch leave, per XCD queue?  (-DRMCV_PROFILE_HANDOVER build)"""
import os, re, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from rmcv_amd import CAMP_BLUE, OPT_PIXEL_GROUPS, STAGE_BINARY, Context, default_params, synth
    n, W, H = 256, 1280, 1024
    torch.cuda.init()
    cs = []
    for k in range(4):      # four contexts with their own frames, taken in turn: every launch reads frames last read four launches ago (HBM, not the Infinity Cache)
        frames = torch.from_numpy(synth.batch(k * 1000003, n, W, H, CAMP_BLUE, 0, threads=16)).cuda()
        c = Context(device=0, max_frames=n, max_width=W, max_height=H)
        c.bind_device_frames(frames.data_ptr(), n, H, W, keepalive=frames)
        c.set_option(OPT_PIXEL_GROUPS, int(sys.argv[2]))
        cs.append(c)
    for rep in range(8):
        print("== rep", rep, flush=True)
        cs[rep % 4].run(default_params(), STAGE_BINARY)
        torch.cuda.synchronize()
    sys.exit(0)
import numpy as np
for g in (2, 3):
    out = subprocess.run([sys.executable, __file__, "child", str(g)], capture_output=True, text=True).stdout
    rep = out.split("== rep 7")[-1]
    kb0 = int(re.search(r"\[kb start\] (\d+)", rep).group(1))
    ev = np.array([[int(a), int(b)] for a, b in re.findall(r"\[kbx\] (\d+) (\d+)", rep)], dtype=np.int64)
    print("groups %d: %d workgroups" % (g, len(ev)))
    for q in range(8):
        t = np.sort((ev[ev[:, 0] == q, 1] - kb0) / 100.0)
        print("   queue %d: first workgroup leaves at %.1f us, median %.1f, last %.1f" % (q, t[0], t[len(t) // 2], t[-1]))
    kb1 = int(re.search(r"\[kb end\] (\d+)", rep).group(1))
    allt = np.sort((ev[:, 1] - kb0) / 100.0)
    print("   kernel: first workgroup leaves at %.1f us, 10 %% have left at %.1f, half at %.1f, 90 %% at %.1f, all at %.1f (end stamp %.1f)" %
          (allt[0], allt[len(allt) // 10], allt[len(allt) // 2], allt[9 * len(allt) // 10], allt[-1], (kb1 - kb0) / 100.0))
