import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from rmcv_amd import Context, CAMP_BLUE, MORPH_NONE
import oracle_lib as O
case = sys.argv[1] if len(sys.argv) > 1 else "dot"
h, w = 64, 64
c = np.zeros((h, w), np.uint8)
if case == "dot": c[10, 10] = 255
elif case == "rect": c[5:20, 8:30] = 255
elif case == "two": c[5:20, 8:30] = 255; c[40:50, 3:9] = 255
elif case == "ring": c[5:40, 5:40] = 255; c[10:35, 10:35] = 0; c[20, 20] = 255
elif case == "noise": c = ((np.random.default_rng(1).random((h, w)) < 0.2) * 255).astype(np.uint8)
img = np.zeros((h, w, 3), np.uint8); img[..., 0] = c
ctx = Context(device=0, max_frames=1, max_width=256, max_height=256)
t = time.time()
pts, offs, b = ctx.extract_color_csr(img, CAMP_BLUE, 80, MORPH_NONE)
rp, ro = O.find_contours(c)
print(case, "ok" if (np.array_equal(pts, rp) and np.array_equal(offs, ro)) else "MISMATCH", len(offs) - 1, "contours", "%.3fs" % (time.time() - t), flush=True)
