# dev tool: per-frame phase times of the sparse kernel (a -DRMCV_PROFILE_HANDOVER build: RMCV_LIB_PATH=rmcv_amd/lib/var_ho.so), one batch alone
import os, sys, re, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from rmcv_amd import CAMP_BLUE, OPT_SPARSE_WAVES, STAGE_ALL, STAGE_BINARY, Context, default_params, synth
    n = 256
    frames = synth.batch(0, n, 1280, 1024, CAMP_BLUE, 0, threads=16)
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    c.set_option(OPT_SPARSE_WAVES, int(sys.argv[2]))
    c.upload(frames)
    for rep in range(3):
        c.run(default_params(), STAGE_BINARY); c.sync()
        print("== rep", rep, flush=True)
        c.run(default_params(), STAGE_ALL & ~STAGE_BINARY); c.sync()
    sys.exit(0)
import numpy as np
for waves in (4, 8):
    out = subprocess.run([sys.executable, __file__, "child", str(waves)], capture_output=True, text=True, timeout=300).stdout
    last = out.split("== rep 2")[-1]
    rows = []
    for ln in last.splitlines():
        m = re.match(r"\[sp\] (\d+) (\d+) (\d+) (\d+) \| ([\d.]+) ([\d.]+) ([\d.]+) ([\d.]+) ([\d.]+) \| elig (\d+)", ln)
        if m:
            g = m.groups()
            rows.append([int(g[0]), (int(g[3]) - int(g[1])) / 100.0] + [float(x) for x in g[4:9]] + [int(g[9])])
    a = np.array(rows)
    if not len(a):
        print("no rows", out[-500:]); continue
    names = ["total", "tables", "cycles", "verify+rank", "fits", "compact+pairs"]
    print("waves %d: %d frames" % (waves, len(a)))
    for k, nm in enumerate(names):
        col = a[:, 1 + k]
        print("  %-14s mean %6.1f  p50 %6.1f  p90 %6.1f  max %6.1f us" % (nm, col.mean(), np.median(col), np.percentile(col, 90), col.max()))
    worst = a[np.argsort(-a[:, 1])[:5]]
    print("  slowest frames (f, total, tables, cycles, verify, fits, pairs, eligible):", [[int(r[0])] + [round(x, 1) for x in r[1:7]] + [int(r[7])] for r in worst])
    print("  eligible contours per frame: mean %.1f max %d" % (a[:, 7].mean(), a[:, 7].max()))
