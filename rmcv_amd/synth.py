"""Synthetic camera stream: deterministic, integer-only frames from csrc/synth.c."""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .abi import CAMP_BLUE, lib, ptr


def frame(index, w=1280, h=1024, camp=CAMP_BLUE, variant=0, out=None):
    """one BGR uint8 frame [h, w, 3]; seed = 20241008 + index"""
    if out is None:
        out = np.empty((h, w, 3), np.uint8)
    rc = lib().rmcv_synth_frame(ptr(out), w, h, 3 * w, C.c_uint64(int(index)), int(camp), int(variant))
    if rc != 0:
        raise ValueError("rmcv_synth_frame failed: %d" % rc)
    return out


def batch(first, n, w=1280, h=1024, camp=CAMP_BLUE, variant=0, threads=8):
    """frames first .. first+n-1 as uint8 [n, h, w, 3] (ctypes releases the GIL, so threads scale)"""
    out = np.empty((n, h, w, 3), np.uint8)
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        list(ex.map(lambda i: frame(first + i, w, h, camp, variant, out[i]), range(n)))
    return out


def checksum(img):
    h, w, _ = img.shape
    img = np.ascontiguousarray(img)
    return int(lib().rmcv_synth_checksum(ptr(img), w, h, 3 * w))


def svm_weights(seed=20241008, n_class=7):
    """stand-in for the reference's svm.xml, which is not in its repository (.gitignore:40-43): seeded weights with
    the trained model's shape (executable/svm/optimizer.cpp:9,16-19 -- 7 classes, linear C_SVC, 20*20*3 features):
    (weights f32 [21, 1200], rho f64 [21], labels i32 [7])"""
    rng = np.random.default_rng(seed)
    n_df = n_class * (n_class - 1) // 2
    w = (rng.standard_normal((n_df, 1200)) * 1e-3).astype(np.float32)
    rho = (rng.standard_normal(n_df) * 0.05).astype(np.float64)
    return w, rho, np.arange(n_class, dtype=np.int32)
