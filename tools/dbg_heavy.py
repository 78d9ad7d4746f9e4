import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ctypes as C
from rmcv_amd import CAMP_BLUE, STAGE_ALL, Pipeline, default_params, synth
n, w, h = 64, 1280, 1024
pl = Pipeline(device=0, max_frames=n, max_width=w, max_height=h, max_contours=4096)
p = default_params()
def mk(kind, one):
    fr = synth.batch(5, n, w, h, CAMP_BLUE, kind, threads=16)
    if one:
        fr[3] = synth.batch(77, 1, w, h, CAMP_BLUE, 14, threads=1)[0]
    return torch.from_numpy(fr).to("cuda:0")
d14, d0one, d0 = mk(14, False), mk(0, True), mk(0, False)
seq = [d0] * 10 + [d14] * 12 + [d0one] * 30 + [d0] * 10
prev = 0
for i, d in enumerate(seq):
    t = pl.submit(d.data_ptr(), n, h, w, p, STAGE_ALL)
    if i >= 3:
        pl.wait(i - 3)
        d_rec, _ = pl.record(i - 3)
        buf = (C.c_uint32 * (n + 3))()
        from rmcv_amd.abi import lib
        lib().rmcv_device_download(0, buf, C.c_void_p(d_rec), C.c_int64(4 * (n + 3)))
        w2 = buf[n + 2]
        info = pl.get_info()
        print(i - 3, "word2 dense", w2 & 0xFFFFF, "points", (w2 >> 20) * 16, "| heavy_batches", info.heavy_batches, "hot", info.hot_batches, "split", info.dense_split)
pl.close()
