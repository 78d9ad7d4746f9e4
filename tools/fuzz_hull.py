# dev tool: cv::minAreaRect on the contours of random scenes, GPU against the oracle.  usage: python tools/fuzz_hull.py [n_scenes] [seed]
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
import oracle_lib as O
from fuzz_contours import random_scene
from rmcv_amd import Context


def main(n=200, seed=9):
    rng = np.random.default_rng(seed)
    c = Context(device=0, max_frames=1, max_width=512, max_height=512, max_contours=8192, max_points=1 << 17)
    O.set_math_mode(0)
    tot = 0
    for t in range(n):
        canvas = random_scene(rng, int(rng.integers(8, 300)), int(rng.integers(8, 400)))
        pts, offs = O.find_contours(canvas)
        for i in range(len(offs) - 1):
            cont = pts[offs[i]:offs[i + 1]]
            g, w = c.min_area_rect(cont), O.min_area_rect(cont)
            if g.tobytes() != w.tobytes():
                print("MISMATCH scene", t, "contour", i, len(cont), g, w)
                return 1
            tot += 1
    print("fuzz ok:", n, "scenes,", tot, "contours")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 9))
