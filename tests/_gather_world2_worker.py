"""worker of tests/test_gpu_gather_world2.py: one rank of a two-rank group ON ONE GPU.  A pipeline with the built-in gather
(rmcv_pipeline_set_gather) on an rmcv_comm whose RCCL is the stand-in of tests/fake_rccl; no torch in this process.
argv: rank world dir nb"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmcv_amd import CAMP_BLUE, STAGE_ALL, Pipeline, abi, default_params, synth  # noqa: E402

rank, world, d, nb = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
L = abi.lib()
n, w, h, depth = 12, 640, 512, 3
idb = (C.c_uint8 * abi.COMM_ID_BYTES)()
assert L.rmcv_comm_unique_id(idb) == 0, "the stand-in librccl.so.1 was not found (LD_LIBRARY_PATH)"
hc = C.c_void_p()
assert L.rmcv_comm_create(idb, world, rank, 0, C.byref(hc)) == 0
pl = Pipeline(device=0, depth=depth, max_frames=n, max_width=w, max_height=h)
pl.set_gather(hc, 0)
sets, dev = [], []
for i in range(nb):                                              # every rank its own frames: (rank, batch) names the stream index
    fr = synth.batch(500000 + 10007 * rank + 131 * i, n, w, h, CAMP_BLUE, i % 2, threads=4)
    p = C.c_void_p()
    assert L.rmcv_device_alloc(0, C.c_int64(fr.nbytes), C.byref(p)) == 0
    assert L.rmcv_device_upload(0, p, abi.ptr(fr), C.c_int64(fr.nbytes)) == 0
    sets.append(fr)
    dev.append(p)
rb = pl.info.record_bytes
own, gathered = [], []
for i in range(nb):
    pl.submit(dev[i].value, n, h, w, default_params(), STAGE_ALL)
    if rank == 1 and i == 2:
        time.sleep(0.3)                                          # the ranks drift apart: the root's receives wait, nothing may overtake
    if i >= depth - 1:
        j = i - (depth - 1)
        arm, offs = pl.collect(j)                                 # (wait covers the gather too)
        own.append((arm, offs))
        if rank == 0:
            dptr, nbytes = pl.gathered(j)
            assert nbytes == world * rb
            buf = np.empty(nbytes, np.uint8)
            assert L.rmcv_device_download(0, abi.ptr(buf), C.c_void_p(dptr), C.c_int64(nbytes)) == 0
            gathered.append(buf)
pl.drain()
for j in range(nb - (depth - 1), nb):
    arm, offs = pl.collect(j)
    own.append((arm, offs))
    if rank == 0:
        dptr, nbytes = pl.gathered(j)
        buf = np.empty(nbytes, np.uint8)
        assert L.rmcv_device_download(0, abi.ptr(buf), C.c_void_p(dptr), C.c_int64(nbytes)) == 0
        gathered.append(buf)
np.savez(os.path.join(d, "rank%d.npz" % rank), armours=np.concatenate([a.view(np.uint8).reshape(-1) for a, _ in own]) if own else np.zeros(0, np.uint8),
         counts=np.array([len(a) for a, _ in own]), offs=np.stack([o for _, o in own]),
         gathered=np.stack(gathered) if gathered else np.zeros((0, 0), np.uint8), record_bytes=rb, armours_offset=pl.info.armours_offset)
pl.close()
L.rmcv_comm_destroy.restype = None
L.rmcv_comm_destroy.argtypes = [C.c_void_p]
L.rmcv_comm_destroy(hc)
print("rank %d done" % rank, flush=True)
