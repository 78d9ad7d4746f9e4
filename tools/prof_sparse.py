# dev tool: phase timings printed by a -DRMCV_PROFILE build of the sparse kernel (RMCV_LIB_PATH=rmcv_amd/lib/var_prof.so)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmcv_amd import CAMP_BLUE, OPT_SPARSE_WAVES, STAGE_ALL, Context, default_params, synth
n = 128
frames = synth.batch(0, n, 1280, 1024, CAMP_BLUE, 0, threads=16)
for waves in (8, 4):
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024)
    c.set_option(OPT_SPARSE_WAVES, waves)
    c.upload(frames)
    print("== waves", waves, flush=True)
    c.run(default_params(), STAGE_ALL)
    c.sync()
    c.close()
