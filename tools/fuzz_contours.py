# dev tool: random compositions of rings / discs / bars / noise -> findContours on the GPU against the oracle; reports how many
# frames stayed on the cycle path.  usage: python tools/fuzz_contours.py [n_images] [seed]
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as O
from rmcv_amd import CAMP_BLUE, MORPH_NONE, Context


def random_scene(rng, h, w):
    img = np.zeros((h, w), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(int(rng.integers(1, 14))):
        kind = int(rng.integers(0, 5))
        cx, cy = int(rng.integers(0, w)), int(rng.integers(0, h))
        a, b = int(rng.integers(2, max(3, w // 3))), int(rng.integers(2, max(3, h // 3)))
        if kind == 0:
            m = (np.abs(xx - cx) <= a) & (np.abs(yy - cy) <= b)
        elif kind == 1:
            m = ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1
        elif kind == 2:   # ring
            t = int(rng.integers(1, 6))
            m = (np.abs(xx - cx) <= a) & (np.abs(yy - cy) <= b) & ~((np.abs(xx - cx) <= a - t) & (np.abs(yy - cy) <= b - t))
        elif kind == 3:   # elliptic ring
            t = rng.uniform(0.4, 0.9)
            r2 = ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2
            m = (r2 <= 1) & (r2 >= t * t)
        else:             # thin line
            th = rng.uniform(0, np.pi)
            d = np.abs((xx - cx) * np.sin(th) - (yy - cy) * np.cos(th))
            m = (d <= rng.uniform(0.4, 1.5)) & (np.abs(xx - cx) <= a) & (np.abs(yy - cy) <= b)
        if rng.random() < 0.25:
            img[m] = 0
        else:
            img[m] = 255
    if rng.random() < 0.3:
        img[rng.random(img.shape) < rng.uniform(0.001, 0.02)] = 255
    if rng.random() < 0.3:
        img[rng.random(img.shape) < rng.uniform(0.001, 0.02)] = 0
    return img


def main(n=500, seed=1):
    rng = np.random.default_rng(seed)
    c = Context(device=0, max_frames=1, max_width=512, max_height=512, max_contours=8192, max_points=1 << 17)
    O.set_math_mode(0)
    fast = 0
    for t in range(n):
        h, w = int(rng.integers(8, 400)), int(rng.integers(8, 500))
        canvas = random_scene(rng, h, w)
        img = np.zeros((h, w, 3), np.uint8)
        img[..., 0] = canvas
        pts, offs, binary = c.extract_color_csr(img, CAMP_BLUE, 80, MORPH_NONE)
        rp, ro = O.find_contours(canvas)
        if not (np.array_equal(binary, canvas) and np.array_equal(offs, ro) and np.array_equal(pts, rp)):
            np.save("gpurun_out/fuzz_fail_%d.npy" % t, canvas)
            print("MISMATCH at image", t, canvas.shape, "contours", len(ro) - 1, len(offs) - 1)
            return 1
        fast += int(c.counts()["status"][0]) == 0
    print("fuzz ok:", n, "images,", fast, "on the cycle path")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 500, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
