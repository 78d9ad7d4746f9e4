"""Host-side mirror of the reference's detection interface over the C-ABI.

Same names, argument meaning and result order as the reference's
  rm::extract_color      include/imgproc.h:29      (src/imgproc.cpp:50-75)
  rm::filter_lightblobs  include/objdetect.h:47-49 (src/objdetect.cpp:55-87)
  rm::filter_armours     include/objdetect.h:70-71 (src/objdetect.cpp:114-166)
plus the batch entry point the north star adds (independent frames resident in HBM).
Every call runs on the GPU through librmcv_hip.so; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import abi
from .abi import (ARMOUR, CAMP_BLUE, LIGHTBLOB, MORPH_CLOSE, POINT, RRECT, STAGE_ALL, LegacyParams, Limits, Params, PnpConfig, RmcvError, default_pnp_config,
                  default_params, lib, ptr)


class Context:
    """One rmcv_ctx: a GPU, its HBM work buffers and a stream.  Single-owner, like the reference's
    process_thread (executable/main.cpp:55)."""

    def __init__(self, device=0, **limits):
        lim = Limits()
        lib().rmcv_default_limits(C.byref(lim))
        for k, v in limits.items():
            setattr(lim, k, v)
        self.limits = lim
        h = C.c_void_p()
        rc = lib().rmcv_ctx_create(int(device), C.byref(lim), C.byref(h))
        if rc != 0:
            raise RmcvError(rc, "rmcv_ctx_create failed (no GPU? this library has no CPU path)")
        self._h = h
        self._made_by = lib()                                     # (a process can hold two builds: bench.py's RMCV_BENCH_AB=lib:...)
        self.device = device
        self._frames_ref = None

    @classmethod
    def borrowed(cls, handle, limits, device=0):
        """a view of a context somebody else owns (a slot of a Pipeline): the same methods, no destroy"""
        self = cls.__new__(cls)
        self.limits, self._h, self._made_by, self.device, self._frames_ref, self._borrowed = limits, C.c_void_p(handle), lib(), device, None, True
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._made_by.rmcv_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RmcvError(rc, lib().rmcv_last_error(self._h).decode())

    # ---------------------------------------------------------------- single frame
    def extract_color(self, image, target=CAMP_BLUE, lower_bound=80, morph=MORPH_CLOSE):
        """rm::extract_color -> (contours, binary); contours = list of (n,2) int32 arrays in
        cv::findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) order"""
        pts, offs, binary = self.extract_color_csr(image, target, lower_bound, morph)
        xy = np.stack([pts["x"], pts["y"]], axis=1) if len(pts) else np.zeros((0, 2), np.int32)
        return [xy[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)], binary

    def extract_color_csr(self, image, target=CAMP_BLUE, lower_bound=80, morph=MORPH_CLOSE):
        image = np.ascontiguousarray(image, np.uint8)
        h, w, ch = image.shape
        assert ch == 3
        binary = np.empty((h, w), np.uint8)
        cap_p, cap_c = self.limits.max_points, self.limits.max_contours
        pts = np.empty(cap_p, POINT)
        offs = np.empty(cap_c + 1, np.int32)
        nc, npnt = C.c_int32(0), C.c_int32(0)
        self._chk(lib().rmcv_extract_color(self._h, ptr(image), w, h, 3 * w, int(target), int(lower_bound), int(morph),
                                           ptr(binary), ptr(pts), cap_p, ptr(offs), cap_c, C.byref(nc), C.byref(npnt)))
        self.shape = (1, h, w)
        return pts[:npnt.value].copy(), offs[:nc.value + 1].copy(), binary

    def filter_lightblobs(self, pts, offs, tilt_max=70.0, ratio_range=(1.5, 80.0), area_range=(10.0, 99999.0),
                          enemy=CAMP_BLUE):
        """rm::filter_lightblobs on CSR contours -> (positive LIGHTBLOB[], source contour index[], negative contour index[])"""
        pts = np.ascontiguousarray(pts, POINT)
        offs = np.ascontiguousarray(offs, np.int32)
        n = len(offs) - 1
        cap = self.limits.max_blobs
        blobs = np.empty(cap, LIGHTBLOB)
        src = np.empty(cap, np.int32)
        neg = np.empty(max(n, 1), np.int32)
        nb, nn = C.c_int32(0), C.c_int32(0)
        self._chk(lib().rmcv_filter_lightblobs(self._h, ptr(pts), ptr(offs), n, C.c_float(tilt_max),
                                               C.c_float(ratio_range[0]), C.c_float(ratio_range[1]),
                                               C.c_double(area_range[0]), C.c_double(area_range[1]), int(enemy),
                                               ptr(blobs), cap, C.byref(nb), ptr(src), ptr(neg), C.byref(nn)))
        return blobs[:nb.value].copy(), src[:nb.value].copy(), neg[:nn.value].copy()

    def filter_armours(self, lightblobs, angle_difference_max=12.0, shear_max=22.0, lenght_ratio_max=0.4, enemy=CAMP_BLUE):
        """rm::filter_armours -> ARMOUR[] in (i,j) lexicographic order"""
        lightblobs = np.ascontiguousarray(lightblobs, LIGHTBLOB)
        cap = self.limits.max_armours
        out = np.empty(cap, ARMOUR)
        na = C.c_int32(0)
        self._chk(lib().rmcv_filter_armours(self._h, ptr(lightblobs), len(lightblobs), C.c_float(angle_difference_max),
                                            C.c_float(shear_max), C.c_float(lenght_ratio_max), int(enemy), ptr(out), cap,
                                            C.byref(na)))
        return out[:na.value].copy()

    def fit_ellipse(self, pts):
        """cv::fitEllipseDirect of one contour (stage-wise parity hook)"""
        pts = np.ascontiguousarray(pts, POINT)
        out = np.zeros(1, RRECT)
        self._chk(lib().rmcv_fit_ellipse(self._h, ptr(pts), len(pts), ptr(out)))
        return out[0]

    def set_option(self, option, value):
        """tuning knobs (abi.OPT_SPARSE_WAVES: 8 = latency of a lone batch, 4 = throughput with several batches in flight)"""
        self._chk(lib().rmcv_ctx_set_option(self._h, int(option), int(value)))

    def check_guards(self):
        """(number of damaged guard zones around the context's device buffers, description of the first) -- 0 in a correct build"""
        n = C.c_int32(-1)
        self._chk(lib().rmcv_ctx_check_guards(self._h, C.byref(n)))
        return n.value, lib().rmcv_last_error(self._h).decode()

    # ---------------------------------------------------------------- armour pose (src/mobility.cpp:166-190, main.cpp:183-192)
    def pnp_load(self, cfg=None):
        """camera matrix, distortion, gripper->camera transform, square size (defaults: the reference's main.cpp literals)"""
        self._pnp = cfg or default_pnp_config()
        self._chk(lib().rmcv_pnp_load(self._h, C.byref(self._pnp)))

    def locate_armours(self, armours, base2gripper=None):
        """per armour: rm::solve_PnP on its vertices + the world transform; returns (rvecs, tvecs, positions), each [n, 3]"""
        arm = np.ascontiguousarray(armours, ARMOUR)
        n = len(arm)
        r, t, p = np.zeros((max(n, 1), 3)), np.zeros((max(n, 1), 3)), np.zeros((max(n, 1), 3))
        b = None if base2gripper is None else np.ascontiguousarray(base2gripper, np.float64).reshape(16)
        self._chk(lib().rmcv_locate_armours(self._h, ptr(arm), n, ptr(b) if b is not None else None, ptr(r), ptr(t), ptr(p)))
        return r[:n], t[:n], p[:n]

    def set_base2gripper(self, mats):
        """one row-major 4x4 per frame of the batch (h_base2gripper, main.cpp:170)"""
        m = np.ascontiguousarray(mats, np.float64).reshape(-1, 16)
        self._chk(lib().rmcv_batch_set_base2gripper(self._h, ptr(m), len(m)))

    def poses(self):
        """(rvecs, tvecs, positions) of all armours of the batch in the order of armours()"""
        cap = self.shape[0] * self.limits.max_armours
        r, t, p = np.zeros((cap, 3)), np.zeros((cap, 3)), np.zeros((cap, 3))
        tot = C.c_int32(0)
        self._chk(lib().rmcv_batch_get_poses(self._h, ptr(r), ptr(t), ptr(p), cap, C.byref(tot)))
        return r[:tot.value].copy(), t[:tot.value].copy(), p[:tot.value].copy()

    # ---------------------------------------------------------------- legacy matcher (src/objdetect.cpp:9-53, 89-112)
    def min_area_rect(self, pts):
        """cv::minAreaRect of one contour (stage-wise parity hook)"""
        pts = np.ascontiguousarray(pts, POINT)
        out = np.zeros(1, RRECT)
        self._chk(lib().rmcv_min_area_rect(self._h, ptr(pts), len(pts), ptr(out)))
        return out[0]

    def match_lightblob(self, contour, min_ratio, max_ratio, tilt_angle, min_area, max_area, fit_ellipse=True):
        """rm::MatchLightBlob: returns (matched, box)"""
        pts = np.ascontiguousarray(contour, POINT)
        lp = LegacyParams(min_ratio, max_ratio, tilt_angle, min_area, max_area, int(bool(fit_ellipse)))
        out = np.zeros(1, RRECT)
        m = C.c_int32(0)
        self._chk(lib().rmcv_match_lightblob(self._h, ptr(pts), len(pts), C.byref(lp), ptr(out), C.byref(m)))
        return bool(m.value), out[0]

    def find_lightblobs(self, pts, offs, min_ratio, max_ratio, tilt_angle, min_area, max_area, source, fit_ellipse=True):
        """rm::FindLightBlobs on CSR contours: returns (blobs, blob_src, boxes); camps are voted from `source` (h, w, 3 BGR)"""
        source = np.ascontiguousarray(source, np.uint8)
        h, w, ch = source.shape
        assert ch == 3
        pts = np.ascontiguousarray(pts, POINT)
        offs = np.ascontiguousarray(offs, np.int32)
        n = len(offs) - 1
        lp = LegacyParams(min_ratio, max_ratio, tilt_angle, min_area, max_area, int(bool(fit_ellipse)))
        cap = max(n, 1)
        blobs, src, boxes = np.zeros(cap, LIGHTBLOB), np.zeros(cap, np.int32), np.zeros(cap, RRECT)
        nb = C.c_int32(0)
        self._chk(lib().rmcv_find_lightblobs(self._h, ptr(source), w, h, 3 * w, ptr(pts), ptr(offs), n, C.byref(lp), ptr(blobs), cap,
                                             C.byref(nb), ptr(src), ptr(boxes)))
        self.shape = (1, h, w)
        return blobs[:nb.value].copy(), src[:nb.value].copy(), boxes[:nb.value].copy()

    @staticmethod
    def lightblob_overlap(blobs, left, right):
        """rm::LightBlobOverlap; raises RmcvError for right == len(blobs), where the reference reads past the end"""
        blobs = np.ascontiguousarray(blobs, LIGHTBLOB)
        o = C.c_int32(0)
        rc = lib().rmcv_lightblob_overlap(ptr(blobs), len(blobs), int(left), int(right), C.byref(o))
        if rc:
            raise RmcvError(rc, "rightIndex == size(): out of range in the reference")
        return bool(o.value)

    @staticmethod
    def max_iou(self_armour, armours):
        """rm::armour::max_IoU: (index, IoU) of the best-overlapping armour, index -1 when none overlaps"""
        me = np.ascontiguousarray(self_armour, ARMOUR).reshape(1)
        arr = np.ascontiguousarray(armours, ARMOUR)
        idx, iou = C.c_int32(0), C.c_float(0)
        rc = lib().rmcv_max_iou(ptr(me), ptr(arr), len(arr), C.byref(idx), C.byref(iou))
        if rc:
            raise RmcvError(rc, "rmcv_max_iou")
        return idx.value, iou.value

    @staticmethod
    def identity_max(history):
        """rm::armour::identity_max over {identity: count}: (identity, probability)"""
        ids = np.array(sorted(history), np.int32)
        cnt = np.array([history[int(k)] for k in ids], np.int32)
        mid, pr = C.c_int32(0), C.c_double(0)
        rc = lib().rmcv_identity_max(ptr(ids), ptr(cnt), len(ids), C.byref(mid), C.byref(pr))
        if rc:
            raise RmcvError(rc, "rmcv_identity_max")
        return mid.value, pr.value

    # ---------------------------------------------------------------- tracker state (src/core.cpp:51-122, main.cpp:57-88)
    @staticmethod
    def track_new(armour, identity, timestamp, position, noise=(5e-5, 0.5, 0.05)):
        """a detected armour as the process loop hands it to the tracking thread (main.cpp:178-195): constructed, identity /
        position / timestamp assigned, reset(5e-5, 0.5, 0.05) -> abi.TRACK record"""
        t = np.zeros(1, abi.TRACK)
        a = np.ascontiguousarray(armour, ARMOUR).reshape(1)
        pos = np.ascontiguousarray(position, np.float64)
        lib().rmcv_track_init(ptr(t), ptr(a), int(identity), C.c_int64(int(timestamp)), ptr(pos))
        if noise is not None:
            lib().rmcv_track_reset(ptr(t), C.c_double(noise[0]), C.c_double(noise[1]), C.c_double(noise[2]))
        return t[0]

    @staticmethod
    def track_update(track, observation, tick_frequency=1e9):
        """rm::armour::update(const armour&): returns the updated abi.TRACK record"""
        t = np.array([track], abi.TRACK)
        o = np.array([observation], abi.TRACK)
        rc = lib().rmcv_track_update(ptr(t), ptr(o), C.c_double(tick_frequency))
        if rc:
            raise RmcvError(rc, "rmcv_track_update")
        return t[0]

    @staticmethod
    def track_predict(track, new_timestamp, tick_frequency=1e9):
        """rm::armour::update(int64)"""
        t = np.array([track], abi.TRACK)
        rc = lib().rmcv_track_predict(ptr(t), C.c_int64(int(new_timestamp)), C.c_double(tick_frequency))
        if rc:
            raise RmcvError(rc, "rmcv_track_predict")
        return t[0]

    @staticmethod
    def track_step(tracking, observations, cap=64, tick_frequency=1e9):
        """one pass of the tracking thread's loop (main.cpp:60-85): returns the new tracking list"""
        buf = np.zeros(cap, abi.TRACK)
        nt = C.c_int32(len(tracking))
        if len(tracking):
            buf[:len(tracking)] = tracking
        obs = np.array(observations, abi.TRACK).copy() if len(observations) else np.zeros(1, abi.TRACK)
        no = C.c_int32(len(observations))
        rc = lib().rmcv_track_step(ptr(buf), C.byref(nt), cap, ptr(obs), C.byref(no), C.c_double(tick_frequency))
        if rc:
            raise RmcvError(rc, "rmcv_track_step")
        return buf[:nt.value].copy()

    def run_legacy(self, legacy, params=None, stages=STAGE_ALL, stream=None):
        """batch path with rm::FindLightBlobs (legacy: LegacyParams) in place of rm::filter_lightblobs"""
        self._params = params or default_params()
        self._legacy = legacy
        self._chk(lib().rmcv_batch_run_legacy(self._h, C.byref(self._params), C.byref(legacy), int(stages), C.c_void_p(stream or 0)))

    # ---------------------------------------------------------------- batch
    def upload(self, frames):
        """frames: uint8 [n, h, w, 3] on the host -> the context's HBM buffer"""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w, ch = frames.shape
        assert ch == 3
        self._chk(lib().rmcv_batch_upload(self._h, ptr(frames), n, w, h, 3 * w, C.c_int64(3 * w * h)))
        self.shape = (n, h, w)

    def bind_device_frames(self, data_ptr, n, h, w, stride=None, frame_pitch=None, keepalive=None):
        """borrow frames already in HBM (e.g. torch_tensor.data_ptr())"""
        stride = stride or 3 * w
        frame_pitch = frame_pitch or stride * h
        self._frames_ref = keepalive
        self._chk(lib().rmcv_batch_set_device_frames(self._h, C.c_void_p(data_ptr), n, w, h, stride, C.c_int64(frame_pitch)))
        self.shape = (n, h, w)

    def run(self, params=None, stages=STAGE_ALL, stream=None):
        self._params = params or default_params()
        self._chk(lib().rmcv_batch_run(self._h, C.byref(self._params), int(stages), C.c_void_p(stream or 0)))

    def run_timed(self, params=None, stages=STAGE_ALL, stream=None):
        """returns ms of [binary, contours, blobs, armours, total] measured with HIP events on the launch stream"""
        self._params = params or default_params()
        ms = (C.c_float * 5)()
        self._chk(lib().rmcv_batch_run_timed(self._h, C.byref(self._params), int(stages), C.c_void_p(stream or 0), ms))
        return [float(x) for x in ms]

    def sync(self):
        self._chk(lib().rmcv_batch_sync(self._h))

    def counts(self):
        n = self.shape[0]
        a = [np.empty(n, np.int32) for _ in range(5)]
        self._chk(lib().rmcv_batch_counts(self._h, *[ptr(x) for x in a]))
        return dict(n_contours=a[0], n_points=a[1], n_blobs=a[2], n_armours=a[3], status=a[4])

    def binary(self, frame):
        n, h, w = self.shape
        out = np.empty((h, w), np.uint8)
        self._chk(lib().rmcv_batch_get_binary(self._h, int(frame), ptr(out)))
        return out

    def contours(self, frame):
        cap_p, cap_c = self.limits.max_points, self.limits.max_contours
        pts = np.empty(cap_p, POINT)
        offs = np.empty(cap_c + 1, np.int32)
        nc, npnt = C.c_int32(0), C.c_int32(0)
        self._chk(lib().rmcv_batch_get_contours(self._h, int(frame), ptr(pts), cap_p, ptr(offs), cap_c, C.byref(nc),
                                                C.byref(npnt)))
        return pts[:npnt.value].copy(), offs[:nc.value + 1].copy()

    def blobs(self, frame):
        cap = self.limits.max_blobs
        blobs = np.empty(cap, LIGHTBLOB)
        src = np.empty(cap, np.int32)
        nb = C.c_int32(0)
        self._chk(lib().rmcv_batch_get_blobs(self._h, int(frame), ptr(blobs), cap, C.byref(nb), ptr(src)))
        return blobs[:nb.value].copy(), src[:nb.value].copy()

    def armours(self):
        """all armours of the batch, frame-major -> (ARMOUR[total], frame_offs[n+1])"""
        n = self.shape[0]
        cap = n * self.limits.max_armours
        out = np.empty(cap, ARMOUR)
        offs = np.empty(n + 1, np.int32)
        tot = C.c_int32(0)
        self._chk(lib().rmcv_batch_get_armours(self._h, ptr(out), cap, ptr(offs), C.byref(tot)))
        return out[:tot.value].copy(), offs

    def device_views(self):
        """(armours device pointer, counts device pointer, per-frame capacity, n_frames) for a collective"""
        a, c = C.c_void_p(), C.c_void_p()
        cap, n = C.c_int32(0), C.c_int32(0)
        self._chk(lib().rmcv_batch_device_views(self._h, C.byref(a), C.byref(c), C.byref(cap), C.byref(n)))
        return a.value, c.value, cap.value, n.value

    def compact_armours_into(self, d_armours_ptr, cap, d_frame_offs_ptr, stream=None):
        """device-side compaction into caller HBM (async on `stream`)"""
        self._chk(lib().rmcv_batch_compact_armours(self._h, C.c_void_p(d_armours_ptr), int(cap), C.c_void_p(d_frame_offs_ptr),
                                                   C.c_void_p(stream or 0)))

    # ---------------------------------------------------------------- icon classifier (BASELINE config 5)
    def svm_load(self, weights, rho, labels):
        """linear one-vs-one C_SVC: weights [n_df, 1200] f32, rho [n_df] f64, labels [n_class] i32"""
        weights = np.ascontiguousarray(weights, np.float32)
        rho = np.ascontiguousarray(rho, np.float64)
        labels = np.ascontiguousarray(labels, np.int32)
        n_class = len(labels)
        assert weights.shape == (n_class * (n_class - 1) // 2, abi.SVM_FEATURES) and len(rho) == weights.shape[0]
        self._chk(lib().rmcv_svm_load(self._h, ptr(weights), ptr(rho), ptr(labels), n_class))

    def classify_armours(self, image, armours):
        """main.cpp:178-181 for one frame -> (identity int32[n], armours with clamped icon, icons uint8[n,20,20,3])"""
        image = np.ascontiguousarray(image, np.uint8)
        h, w, _ = image.shape
        arm = np.ascontiguousarray(armours, ARMOUR).copy()
        n = len(arm)
        ident = np.empty(max(n, 1), np.int32)
        icons = np.empty((max(n, 1), 20, 20, 3), np.uint8)
        self._chk(lib().rmcv_classify_armours(self._h, ptr(image), w, h, 3 * w, ptr(arm), n, ptr(ident), ptr(icons)))
        return ident[:n].copy(), arm, icons[:n].copy()

    def identities(self):
        n = self.shape[0]
        cap = n * self.limits.max_armours
        out = np.empty(cap, np.int32)
        tot = C.c_int32(0)
        self._chk(lib().rmcv_batch_get_identities(self._h, ptr(out), cap, C.byref(tot)))
        return out[:tot.value].copy()

    def icons(self, frame):
        cap = self.limits.max_armours
        out = np.empty((cap, 20, 20, 3), np.uint8)
        na = C.c_int32(0)
        self._chk(lib().rmcv_batch_get_icons(self._h, int(frame), ptr(out), cap, C.byref(na)))
        return out[:na.value].copy()

    def detect_batch(self, frames, params=None):
        """the whole path of executable/main.cpp:172-176 on a batch of host frames"""
        self.upload(frames)
        self.run(params)
        self.sync()
        return self.armours()
