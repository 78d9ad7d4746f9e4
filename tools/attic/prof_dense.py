# dev tool: phase timings of the mid tier and the fused tail on dense frames (-DRMCV_PROFILE build: RMCV_LIB_PATH=rmcv_amd/lib/var_profile.so)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmcv_amd import CAMP_BLUE, OPT_SPARSE_WAVES, STAGE_ALL, Context, default_params, synth
n = 8
for level in (2, 4):
    frames = synth.batch(0, n, 1280, 1024, CAMP_BLUE, 10 + level, threads=8)
    for waves in (4, 8):
        c = Context(device=0, max_frames=n, max_width=1280, max_height=1024, max_contours=4096)
        c.set_option(OPT_SPARSE_WAVES, waves)
        c.upload(frames)
        for rep in range(2):
            print("== dense level", level, "waves", waves, "rep", rep, flush=True)
            c.run(default_params(), STAGE_ALL)
            c.sync()
        c.close()
