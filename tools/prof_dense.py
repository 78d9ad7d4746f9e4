# dev tool: phase timings of the sparse kernel on dense frames (a -DRMCV_PROFILE build of the 4-wavefront kernel:
#   bash tools/build_variant.sh prof k_contours_w4.hip "-DRMCV_PROFILE";  RMCV_LIB_PATH=rmcv_amd/lib/var_prof.so python tools/prof_dense.py)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rmcv_amd import CAMP_BLUE, OPT_SPARSE_WAVES, STAGE_ALL, STAGE_BINARY, Context, default_params, synth
n = 64
for var in (12, 13, 14):
    frames = synth.batch(0, n, 1280, 1024, CAMP_BLUE, var, threads=16)
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024, max_contours=4096)
    c.set_option(OPT_SPARSE_WAVES, 4)
    c.upload(frames)
    print("== variant", var, flush=True)
    c.run(default_params(), STAGE_BINARY)
    c.sync()
    t0 = time.perf_counter()
    c.run(default_params(), STAGE_ALL & ~STAGE_BINARY)
    c.sync()
    print("   sparse stage of %d frames alone: %.3f ms" % (n, (time.perf_counter() - t0) * 1e3), flush=True)
    cnt = c.counts()
    print("   contours/frame %.0f points/frame %.0f blobs/frame %.1f mid %d" % (cnt["n_contours"].mean(), cnt["n_points"].mean(), cnt["n_blobs"].mean(), int(np.count_nonzero(cnt["status"] & 64))), flush=True)
    c.close()
