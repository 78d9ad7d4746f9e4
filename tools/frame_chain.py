"""dev tool: the per-frame drop-in chain (rmcv_extract_color -> rmcv_filter_lightblobs -> rmcv_filter_armours on ONE host frame, results
back on the host after every call) timed alone: median / min of 200 chains per upload mode.   python tools/frame_chain.py [W H]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmcv_amd import CAMP_BLUE, MORPH_CLOSE, OPT_FRAME_UPLOAD, Context, synth
from rmcv_amd.abi import ARMOUR, LIGHTBLOB, POINT, lib, ptr

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1280, 1024)
L = lib()
c1 = Context(device=0, max_frames=1, max_width=W, max_height=H)
imgs = [np.ascontiguousarray(synth.frame(i, W, H, CAMP_BLUE, 0)) for i in range(4)]
binary = np.empty((H, W), np.uint8)
pts, offs = np.empty(c1.limits.max_points, POINT), np.empty(c1.limits.max_contours + 1, np.int32)
blobs, neg = np.empty(c1.limits.max_blobs, LIGHTBLOB), np.empty(c1.limits.max_contours, np.int32)
arms = np.empty(c1.limits.max_armours, ARMOUR)
nc, npt, nb, nn, na = (C.c_int32(0) for _ in range(5))


def one_frame(img):
    t = [time.perf_counter()]
    rc = L.rmcv_extract_color(c1._h, ptr(img), W, H, 3 * W, CAMP_BLUE, 80, MORPH_CLOSE, ptr(binary), ptr(pts), len(pts), ptr(offs),
                              len(offs) - 1, C.byref(nc), C.byref(npt))
    t.append(time.perf_counter())
    rc |= L.rmcv_filter_lightblobs(c1._h, ptr(pts), ptr(offs), nc.value, C.c_float(70.0), C.c_float(1.5), C.c_float(80.0), C.c_double(10.0),
                                   C.c_double(99999.0), CAMP_BLUE, ptr(blobs), len(blobs), C.byref(nb), None, ptr(neg), C.byref(nn))
    t.append(time.perf_counter())
    rc |= L.rmcv_filter_armours(c1._h, ptr(blobs), nb.value, C.c_float(12.0), C.c_float(22.0), C.c_float(0.4), CAMP_BLUE, ptr(arms), len(arms),
                                C.byref(na))
    t.append(time.perf_counter())
    assert rc == 0
    return [(t[i + 1] - t[i]) * 1e3 for i in range(3)]


for mode, name in ((0, "runtime_pageable"), (2, "registered_in_place"), (1, "pinned_staging")):
    c1.set_option(OPT_FRAME_UPLOAD, mode)
    for i in range(8):
        one_frame(imgs[i % 4])
    ts = np.array([one_frame(imgs[i % 4]) for i in range(200)])
    tot = ts.sum(1)
    print("%-20s median %.4f  min %.4f ms | extract_color %.4f  filter_lightblobs %.4f  filter_armours %.4f | armours %d" %
          (name, np.median(tot), tot.min(), np.median(ts[:, 0]), np.median(ts[:, 1]), np.median(ts[:, 2]), na.value))
