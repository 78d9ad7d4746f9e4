"""rmcv_pipeline_* (include/rmcv_abi.h): the pipelined batch schedule, owned by the library.

The reference's process loop (executable/main.cpp:163-209) takes a frame, detects, hands the armours on.  Its batch form keeps
`depth` batches in flight on one GPU -- pixel kernels of consecutive batches alternating over two streams, the per-frame kernels on
four higher-priority streams, everything chained by events inside librmcv_hip.so.  This module is the thin ctypes face of the three
calls (submit / collect / drain) plus the hook through which a device-side consumer -- the multi-GPU gather -- rides along.
"""
import ctypes as C

import numpy as np

from . import abi
from .abi import ARMOUR, STAGE_ALL, Limits, PipelineConfig, PipelineInfo, RmcvError, default_params, lib, ptr
from .api import Context


class Pipeline:
    def __init__(self, device=0, depth=0, pixel_streams=0, sparse_streams=0, armour_cap=0, sparse_waves=0, pixel_groups=0,
                 host_results=0, dense_streams=0, hot_contexts=0, **limits):
        lim = Limits()
        lib().rmcv_default_limits(C.byref(lim))
        for k, v in limits.items():
            setattr(lim, k, v)
        self.limits = lim
        cfg = PipelineConfig(depth, pixel_streams, sparse_streams, armour_cap, sparse_waves, pixel_groups, host_results, dense_streams, hot_contexts, 0)
        h = C.c_void_p()
        rc = lib().rmcv_pipeline_create(int(device), C.byref(lim), C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RmcvError(rc, "rmcv_pipeline_create failed (no GPU? this library has no CPU path)")
        self._h, self._lib, self.device = h, lib(), device
        self.info = self.get_info()
        self.depth = self.info.depth
        self._hook = None              # keeps the ctypes callback alive
        self._keep = {}                # slot -> the frames object of the batch in flight there
        self._n = {}                   # slot -> frames of the batch that lives there
        self._ticket = C.c_uint64(0)
        self._params = default_params()
        self.contexts = [Context.borrowed(self._lib.rmcv_pipeline_context(self._h, k), lim, device) for k in range(self.depth)]

    def close(self):
        if getattr(self, "_h", None):
            for c in self.contexts:
                c.close()
            self._lib.rmcv_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RmcvError(rc, self._lib.rmcv_pipeline_last_error(self._h).decode())

    def get_info(self):
        o = PipelineInfo()
        self._chk(self._lib.rmcv_pipeline_get_info(self._h, C.byref(o)))
        return o

    # ------------------------------------------------------------------ the three calls
    def submit(self, data_ptr, n, h, w, params=None, stages=STAGE_ALL, stride=None, frame_pitch=None, keepalive=None, legacy=None):
        """enqueue one batch of frames resident in HBM (data_ptr: e.g. torch_tensor.data_ptr()); returns the ticket"""
        if params is not None:
            self._params = params
        stride = stride or 3 * w
        frame_pitch = frame_pitch or stride * h
        if legacy is not None:
            rc = self._lib.rmcv_pipeline_submit_legacy(self._h, data_ptr, n, w, h, stride, frame_pitch, C.addressof(self._params), C.addressof(legacy),
                                                       int(stages), C.addressof(self._ticket))
        else:
            rc = self._lib.rmcv_pipeline_submit(self._h, data_ptr, n, w, h, stride, frame_pitch, C.addressof(self._params), int(stages),
                                                C.addressof(self._ticket))
        if rc != 0:
            self._chk(rc)
        t = self._ticket.value
        self._keep[t % self.depth] = keepalive
        self._n[t % self.depth] = n
        self.shape = (n, h, w)
        for c in self.contexts:
            c.shape = self.shape
        return t

    def wait(self, ticket):
        self._chk(self._lib.rmcv_pipeline_wait(self._h, int(ticket)))

    def collect(self, ticket, cap=None):
        """(ARMOUR[total], frame_offs int32[n + 1]) of the batch, frame-major"""
        cap = cap or self.info.armour_cap
        out = np.empty(cap, ARMOUR)
        offs = np.empty(self.limits.max_frames + 1, np.int32)
        tot = C.c_int32(0)
        self._chk(self._lib.rmcv_pipeline_collect(self._h, int(ticket), ptr(out), cap, ptr(offs), C.addressof(tot)))
        n = self._n[int(ticket) % self.depth]
        return out[:tot.value].copy(), offs[:n + 1].copy()

    def set_hot_contexts(self, n):
        """rmcv_pipeline_config::hot_contexts from the next submit on (0: off)"""
        self._chk(self._lib.rmcv_pipeline_set_hot_contexts(self._h, int(n)))

    def set_wait_timeout(self, ms):
        """deadline of wait / collect / drain in milliseconds (0: none); a wait that runs out raises RmcvError(ERR_TIMEOUT)"""
        self._chk(self._lib.rmcv_pipeline_set_wait_timeout(self._h, int(ms)))

    def drain(self):
        self._chk(self._lib.rmcv_pipeline_drain(self._h))

    def context_of(self, ticket):
        """the Context view of the slot a (waited-for) ticket lives in: per-stage getters (binary, contours, blobs, counts)"""
        h = self._lib.rmcv_pipeline_context_of(self._h, int(ticket))
        if not h:
            raise RmcvError(abi.ERR_BAD_ARG, "ticket %d: %s" % (ticket, self._lib.rmcv_pipeline_last_error(self._h).decode()))
        for c in self.contexts:
            if c._h.value == h:
                return c
        raise RmcvError(abi.ERR_BAD_ARG, "unknown context")

    def record(self, ticket):
        """(device pointer of the ticket's record, hipStream_t it is produced on) as ints"""
        d, s = C.c_void_p(), C.c_void_p()
        self._chk(self._lib.rmcv_pipeline_record(self._h, int(ticket), C.addressof(d), C.addressof(s)))
        return d.value, s.value or 0

    # ------------------------------------------------------------------ device-side consumers
    def set_hook(self, fn):
        """fn(ticket, d_record: int, record_bytes: int, hip_stream: int) -> None | a hipEvent_t (int) recorded when the record has been
        read on another stream; called on the submitting thread right behind the enqueue of every batch's compaction"""
        if fn is None:
            self._chk(self._lib.rmcv_pipeline_set_hook(self._h, None, None))
            self._hook = None
            return

        def tramp(_user, ticket, d_record, record_bytes, hip_stream, done_event):
            try:
                ev = fn(int(ticket), int(d_record or 0), int(record_bytes), int(hip_stream or 0))
                if ev:
                    done_event[0] = ev
                return 0
            except Exception:                                    # noqa: BLE001 -- never unwind through the C frame
                import traceback
                traceback.print_exc()
                return abi.ERR_HIP
        self._hook = abi.PIPELINE_HOOK(tramp)
        self._chk(self._lib.rmcv_pipeline_set_hook(self._h, self._hook, None))

    def set_gather(self, comm_handle, root=0):
        """built-in hook: rmcv_gather of every batch's record on an rmcv_comm (rmcv_amd.dist.AbiGather._h)"""
        self._chk(self._lib.rmcv_pipeline_set_gather(self._h, comm_handle, int(root)))

    def gathered(self, ticket):
        """root: (device pointer of n_ranks x record_bytes, bytes); others: (None, bytes)"""
        d, b = C.c_void_p(), C.c_int64(0)
        self._chk(self._lib.rmcv_pipeline_gathered(self._h, int(ticket), C.addressof(d), C.addressof(b)))
        return d.value, b.value
