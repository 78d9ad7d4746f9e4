# dev tool: the full path (extract_color -> filter_lightblobs -> filter_armours) on random scenes, GPU against the oracle.
# usage: python tools/fuzz_path.py [n_batches] [seed]
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
import oracle_lib as O
from fuzz_contours import random_scene
from rmcv_amd import CAMP_BLUE, STAGE_ALL, Context, default_params


def main(nb=20, seed=3, loose=0):
    rng = np.random.default_rng(seed)
    h, w, n = 256, 320, 32
    c = Context(device=0, max_frames=n, max_width=w, max_height=h, max_contours=4096, max_points=1 << 16, max_blobs=1024, max_armours=4096)
    O.set_math_mode(0)
    p = O.default_params()
    gp = default_params()
    if loose:   # every gate wide open: every contour with >= 6 points is fitted, nearly every pair of blobs becomes an armour
        for q in (p, gp):
            q.tilt_max, q.ratio_lo, q.ratio_hi, q.area_lo, q.area_hi = 1e9, 0.0, 1e30, 0.0, 1e30
            q.angle_diff_max, q.shear_max, q.length_ratio_max = 1e9, 1e9, 0.0
    tot_b = tot_a = 0
    for b in range(nb):
        frames = np.zeros((n, h, w, 3), np.uint8)
        for f in range(n):
            frames[f, ..., 0] = random_scene(rng, h, w)
            frames[f, ..., 2] = rng.integers(0, 60, (h, w), dtype=np.uint8)
        c.upload(frames)
        c.run(gp, STAGE_ALL)
        c.sync()
        arm, offs = c.armours()
        st = c.counts()["status"]
        for f in range(n):
            if st[f] & 15:
                continue                                        # a capacity of this small context was exceeded: reported, not compared
            ref = O.detect_frame(frames[f], p, cap_blobs=4096, cap_armours=1 << 16)
            blobs, _ = c.blobs(f)
            a = arm[offs[f]:offs[f + 1]]
            if blobs.tobytes() != ref["blobs"].tobytes() or a.tobytes() != ref["armours"].tobytes():
                np.save("gpurun_out/fuzz_path_fail_%d_%d.npy" % (b, f), frames[f])
                print("MISMATCH batch", b, "frame", f, "blobs", len(blobs), len(ref["blobs"]), "armours", len(a), len(ref["armours"]))
                return 1
            tot_b += len(blobs)
            tot_a += len(a)
    print("fuzz ok:", nb * n, "scenes,", tot_b, "light blobs,", tot_a, "armours")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 3,
                  int(sys.argv[3]) if len(sys.argv) > 3 else 0))
