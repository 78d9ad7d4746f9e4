"""Prototype (CPU, numpy/python) of the cycle formulation of cv::findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) that a parallel
device version could use: every visit of the border follower to a pixel is a node (pixel, maximal arc of background neighbours
that contains a 4-neighbour), the follower's step is a bijection on nodes, a contour is a cycle, its start is the cycle's
raster-first node and it is an outer border iff that node's arc contains the west neighbour.  Checked against the oracle."""
import sys, os
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle_lib as O

DX = [1, 1, 0, -1, -1, -1, 0, 1]   # 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE  (y down)
DY = [0, -1, -1, -1, 0, 1, 1, 1]


def arcs_of_ring(ring):
    """ring: list of 8 bools (fg) by direction.  returns list of (back_dir, next_dir, has_west) for every real visit, the arc that
    contains W first.  back_dir = fg neighbour at the clockwise end of the arc, next_dir = fg neighbour at its ccw end."""
    if not any(ring):
        return [(-1, -1, True)]                      # isolated pixel: one visit, stays
    out = []
    for b in range(8):
        if not ring[b]:
            continue
        # arc = directions b+1, b+2, ... while background
        arc = []
        d = (b + 1) & 7
        while not ring[d]:
            arc.append(d)
            d = (d + 1) & 7
        if not arc:
            continue
        if not any(a in (0, 2, 4, 6) for a in arc):
            continue                                  # touches the background region only diagonally: not a visit
        out.append((b, d, 4 in arc))
    out.sort(key=lambda t: (not t[2], t[0]))
    return out


def find_contours_cycles(img):
    h, w = img.shape
    f = np.zeros((h + 2, w + 2), bool)
    f[1:-1, 1:-1] = img != 0
    nodes = []           # (y, x, back, next_dir, has_west)
    index = {}           # (y, x, back) -> node id
    for y in range(1, h + 1):
        for x in range(1, w + 1):
            if not f[y, x]:
                continue
            ring = [bool(f[y + DY[d], x + DX[d]]) for d in range(8)]
            if all(ring):
                continue
            for (b, nd, hw) in arcs_of_ring(ring):
                index[(y, x, b)] = len(nodes)
                nodes.append((y, x, b, nd, hw))
    n = len(nodes)
    nxt = np.zeros(n, np.int64)
    for i, (y, x, b, nd, hw) in enumerate(nodes):
        if b < 0:
            nxt[i] = i
        else:
            nxt[i] = index[(y + DY[nd], x + DX[nd], (nd + 4) & 7)]
    assert len(set(nxt.tolist())) == n, "the step is not a bijection"
    seen = np.zeros(n, bool)
    contours = []
    for i in range(n):                               # node ids are in raster order, W arc first within a pixel
        if seen[i]:
            continue
        cyc = []
        j = i
        while not seen[j]:
            seen[j] = True
            cyc.append(j)
            j = nxt[j]
        assert j == i
        if nodes[i][4]:                              # the raster-first node's arc contains W: an outer border
            contours.append([(nodes[k][1] - 1, nodes[k][0] - 1) for k in cyc])
    return contours[::-1]                            # findContours returns the last found first


def check(img):
    pts, offs = O.find_contours(img)
    want = O.contours_as_lists(pts, offs)
    got = find_contours_cycles(img)
    return want, got


if __name__ == "__main__":
    from rmcv_amd import synth
    rng = np.random.default_rng(0)
    bad = 0
    total = 0
    for t in range(300):
        hh, ww = int(rng.integers(3, 40)), int(rng.integers(3, 48))
        img = (rng.random((hh, ww)) < rng.uniform(0.05, 0.6)).astype(np.uint8) * 255
        want, got = check(img)
        # nested components (inside holes) are dropped by RETR_EXTERNAL but not by this prototype: compare as sets/order of the common ones
        if want != got:
            ws, gs = set(map(tuple, want)), set(map(tuple, got))
            if not ws <= gs or [c for c in got if tuple(c) in ws] != want:
                bad += 1
                print("MISMATCH", t, img.shape, len(want), len(got))
        total += 1
    print("random images:", total, "bad", bad)
    for i in range(6):
        fr = synth.frame(i, 1280, 1024, 1, i & 1)
        b = O.extract_binary(fr)
        ys, xs = np.nonzero(b)
        # crop to keep the python loops cheap
        want, got = check(b)
        print("frame", i, "contours", len(want), "equal", want == got, "nodes~", sum(len(c) for c in got))
