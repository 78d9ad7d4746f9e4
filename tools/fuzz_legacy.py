# dev tool: rm::FindLightBlobs (both box kinds, camp vote) on random coloured scenes, GPU against the oracle.
# usage: python tools/fuzz_legacy.py [n_scenes] [seed]
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
import oracle_lib as O
from fuzz_contours import random_scene
from rmcv_amd import Context


def main(n=200, seed=4):
    rng = np.random.default_rng(seed)
    c = Context(device=0, max_frames=1, max_width=512, max_height=512, max_contours=8192, max_points=1 << 17, max_blobs=8192)
    O.set_math_mode(0)
    tot = 0
    for t in range(n):
        h, w = int(rng.integers(16, 300)), int(rng.integers(16, 400))
        canvas = random_scene(rng, h, w)
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)             # the camp vote reads the colours under each bounding box
        img[canvas != 0] = np.maximum(img[canvas != 0], rng.integers(0, 256, 3, dtype=np.uint8))
        pts, offs = O.find_contours(canvas)
        if len(offs) - 1 > 4000:
            continue
        for fit in (True, False):
            args = (float(rng.uniform(1.0, 2.0)), float(rng.uniform(3, 100)), float(rng.uniform(10, 180)), float(rng.uniform(0, 20)), 1e9)
            gb, gs, gx = c.find_lightblobs(pts, offs, *args, img, fit)
            ob, os_, ox = O.find_lightblobs(img, pts, offs, *args, fit)
            if not (np.array_equal(gs, os_) and gx.tobytes() == ox.tobytes() and gb.tobytes() == ob.tobytes()):
                print("MISMATCH scene", t, "fit", fit, len(gb), len(ob))
                np.save("gpurun_out/fuzz_legacy_fail_%d.npy" % t, canvas)
                return 1
            tot += len(ob)
    print("fuzz ok:", n, "scenes,", tot, "light blobs")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 4))
