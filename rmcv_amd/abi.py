"""ctypes binding of librmcv_hip.so (the C-ABI declared in include/rmcv_abi.h).

There is no fallback: if the HIP library is missing or no GPU is usable, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RMCV_LIB_PATH") or os.path.join(_HERE, "lib", "librmcv_hip.so")  # env: dev A/B builds

# PODs of include/rmcv_abi.h as numpy dtypes
POINT = np.dtype([("x", "<i4"), ("y", "<i4")])
RRECT = np.dtype([("cx", "<f4"), ("cy", "<f4"), ("w", "<f4"), ("h", "<f4"), ("angle", "<f4")])
LIGHTBLOB = np.dtype([("angle", "<f4"), ("target", "<i4"), ("center", "<f4", (2,)),
                      ("vertices", "<f4", (4, 2)), ("size", "<f4", (2,))])
ARMOUR = np.dtype([("icon", "<f4", (4, 2)), ("vertices", "<f4", (4, 2)), ("bbox", "<f4", (4,)),
                   ("blob_i", "<i4"), ("blob_j", "<i4")])
TRACK_IDS = 32
TRACK = np.dtype([("armour", ARMOUR), ("timestamp", "<i8"), ("lost_count", "<i4"), ("identity", "<i4"), ("position", "<f8", (3,)),
                  ("initialized", "<i4"), ("n_ids", "<i4"), ("ids", "<i4", (TRACK_IDS,)), ("counts", "<i4", (TRACK_IDS,)),
                  ("measurement", "<f8", (6,)), ("state_pre", "<f8", (6,)), ("state_post", "<f8", (6,)),
                  ("transition", "<f8", (6, 6)), ("measurement_matrix", "<f8", (6, 6)), ("process_noise_cov", "<f8", (6, 6)),
                  ("measurement_noise_cov", "<f8", (6, 6)), ("error_cov_pre", "<f8", (6, 6)), ("error_cov_post", "<f8", (6, 6)),
                  ("gain", "<f8", (6, 6))])  # rmcv_track
assert POINT.itemsize == 8 and LIGHTBLOB.itemsize == 56 and ARMOUR.itemsize == 88 and TRACK.itemsize == 2552

OK, ERR_BAD_ARG, ERR_CAPACITY, ERR_NOMEM, ERR_HIP, ERR_NO_DEVICE, ERR_RCCL, ERR_TIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7
COMM_ID_BYTES = 128
CAMP_RED, CAMP_BLUE, CAMP_GUIDELIGHT, CAMP_NEUTRAL = 0, 1, 2, -1
MORPH_NONE, MORPH_DILATE, MORPH_CLOSE = 0, 1, 2
OPT_SPARSE_WAVES = 1
OPT_PIXEL_GROUPS = 2
OPT_FRAME_UPLOAD = 3
OPT_RUN_AHEAD = 4
OPT_CONTOUR_TIER = 5
OPT_DENSE_DEFER = 7
OPT_PIXEL_HALO_NT = 11
OPT_OVERLOADS = 13
OPT_PIXEL_SHAPE = 14
OPT_WAIT_TIMEOUT_MS = 15
OPT_TEST_DELAY_US = 16
OPT_IMAGE_EXPORT = 17
OPT_TEST_SLOW_US = 18
STAGE_BINARY, STAGE_CONTOURS, STAGE_BLOBS, STAGE_ARMOURS, STAGE_ALL, STAGE_IDENTITY, STAGE_POSE, STAGE_NO_IMAGE = 1, 2, 4, 8, 15, 16, 32, 64
SVM_FEATURES = 1200
FRAME_OVF_CONTOURS, FRAME_OVF_POINTS, FRAME_OVF_BLOBS, FRAME_OVF_ARMOURS, FRAME_SLOW_PATH, FRAME_MID_PATH = 1, 2, 4, 8, 16, 64

EXPORTS = [
    "rmcv_abi_version", "rmcv_default_params", "rmcv_default_limits", "rmcv_ctx_create", "rmcv_ctx_destroy",
    "rmcv_last_error", "rmcv_ctx_set_option", "rmcv_ctx_forget_frame_buffer", "rmcv_ctx_check_guards", "rmcv_ctx_frame_timing", "rmcv_extract_color", "rmcv_filter_lightblobs", "rmcv_filter_armours", "rmcv_fit_ellipse",
    "rmcv_batch_upload", "rmcv_batch_set_device_frames", "rmcv_batch_run", "rmcv_batch_sync", "rmcv_batch_run_timed",
    "rmcv_batch_counts", "rmcv_batch_get_binary", "rmcv_batch_get_contours", "rmcv_batch_get_blobs",
    "rmcv_batch_get_armours", "rmcv_batch_device_views", "rmcv_batch_compact_armours", "rmcv_synth_frame", "rmcv_synth_checksum",
    "rmcv_svm_load", "rmcv_classify_armours", "rmcv_batch_get_identities", "rmcv_batch_get_icons",
    "rmcv_default_pnp_config", "rmcv_pnp_load", "rmcv_locate_armours", "rmcv_batch_set_base2gripper", "rmcv_batch_get_poses",
    "rmcv_max_iou", "rmcv_identity_max", "rmcv_comm_unique_id", "rmcv_comm_create", "rmcv_comm_destroy", "rmcv_comm_info", "rmcv_comm_last_error", "rmcv_gather",
    "rmcv_default_pipeline_config", "rmcv_pipeline_create", "rmcv_pipeline_destroy", "rmcv_pipeline_last_error", "rmcv_pipeline_get_info", "rmcv_pipeline_context", "rmcv_pipeline_context_of", "rmcv_pipeline_set_hot_contexts", "rmcv_pipeline_set_wait_timeout", "rmcv_pipeline_reset_stats", "rmcv_hw_queues_hint", "rmcv_pixel_ws_launches",
    "rmcv_pipeline_submit", "rmcv_pipeline_submit_legacy", "rmcv_pipeline_wait", "rmcv_pipeline_collect", "rmcv_pipeline_drain", "rmcv_pipeline_record",
    "rmcv_pipeline_set_hook", "rmcv_pipeline_set_gather", "rmcv_pipeline_gathered", "rmcv_device_alloc", "rmcv_device_free", "rmcv_device_upload", "rmcv_device_download",
    "rmcv_track_init", "rmcv_track_reset", "rmcv_track_update", "rmcv_track_predict", "rmcv_track_step", "rmcv_min_area_rect", "rmcv_match_lightblob", "rmcv_find_lightblobs", "rmcv_lightblob_overlap", "rmcv_batch_run_legacy",
]


class Params(C.Structure):
    """rmcv_params; defaults are the literals of the reference's executable/main.cpp:172-176"""
    _fields_ = [("camp", C.c_int32), ("lower_bound", C.c_int32), ("morph", C.c_int32), ("tilt_max", C.c_float),
                ("ratio_lo", C.c_float), ("ratio_hi", C.c_float), ("area_lo", C.c_double), ("area_hi", C.c_double),
                ("angle_diff_max", C.c_float), ("shear_max", C.c_float), ("length_ratio_max", C.c_float),
                ("_pad", C.c_int32)]


class LegacyParams(C.Structure):
    """rmcv_legacy_params: the float arguments of rm::MatchLightBlob / rm::FindLightBlobs (include/objdetect.h:22-37)"""
    _fields_ = [("min_ratio", C.c_float), ("max_ratio", C.c_float), ("tilt_angle", C.c_float), ("min_area", C.c_float),
                ("max_area", C.c_float), ("fit_ellipse", C.c_int32)]


class PnpConfig(C.Structure):
    """rmcv_pnp_config: cammat / discof / h_gripper2camera / exactSize of the reference's executable/main.cpp:7-19, 184"""
    _fields_ = [("camera_matrix", C.c_double * 9), ("dist", C.c_double * 5), ("gripper2camera", C.c_double * 16),
                ("square_w", C.c_float), ("square_h", C.c_float)]


class Limits(C.Structure):
    _fields_ = [("max_frames", C.c_int32), ("max_width", C.c_int32), ("max_height", C.c_int32),
                ("max_contours", C.c_int32), ("max_points", C.c_int32), ("max_blobs", C.c_int32),
                ("max_armours", C.c_int32), ("_pad", C.c_int32)]


class PipelineConfig(C.Structure):
    """rmcv_pipeline_config (0 in a field = the default)"""
    _fields_ = [("depth", C.c_int32), ("pixel_streams", C.c_int32), ("sparse_streams", C.c_int32), ("armour_cap", C.c_int32),
                ("sparse_waves", C.c_int32), ("pixel_groups", C.c_int32), ("host_results", C.c_int32), ("dense_streams", C.c_int32),
                ("hot_contexts", C.c_int32), ("_reserved", C.c_int32)]


class PipelineInfo(C.Structure):
    """rmcv_pipeline_info"""
    _fields_ = [("depth", C.c_int32), ("pixel_streams", C.c_int32), ("sparse_streams", C.c_int32), ("armour_cap", C.c_int32),
                ("sparse_waves", C.c_int32), ("pixel_groups", C.c_int32), ("host_results", C.c_int32), ("dense_streams", C.c_int32),
                ("max_frames", C.c_int32), ("hw_queues_env", C.c_int32), ("hw_queues_wanted", C.c_int32), ("_pad", C.c_int32),
                ("record_bytes", C.c_int64), ("armours_offset", C.c_int64), ("submitted", C.c_uint64), ("collected", C.c_uint64),
                ("dense_split", C.c_uint64), ("hot_batches", C.c_uint64), ("hot_contexts", C.c_int32), ("_pad2", C.c_int32), ("latency_batches", C.c_uint64),
                ("host_blocking_calls", C.c_uint64), ("wait_timeout_ms", C.c_int32), ("_pad3", C.c_int32), ("max_submit_us", C.c_double), ("heavy_batches", C.c_uint64), ("held_back", C.c_uint64)]


# rmcv_pipeline_hook: int (*)(void* user, uint64_t ticket, void* d_record, int64_t record_bytes, void* hip_stream, void** done_event)
PIPELINE_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p))


class RmcvError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("rmcv error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load(path):
    """a configured handle of one build of the library (not installed as THE library: see use)"""
    if not os.path.exists(path):
        raise ImportError("%s is missing: run `make -C rmcv_amd/csrc` (or __graft_entry__.build())" % path)
    L = C.CDLL(path)
    L.rmcv_last_error.restype = C.c_char_p
    L.rmcv_last_error.argtypes = [C.c_void_p]
    L.rmcv_synth_checksum.restype = C.c_uint64
    L.rmcv_ctx_destroy.restype = None
    L.rmcv_ctx_destroy.argtypes = [C.c_void_p]
    L.rmcv_pipeline_last_error.restype = C.c_char_p
    L.rmcv_pipeline_last_error.argtypes = [C.c_void_p]
    L.rmcv_pipeline_destroy.restype = None
    L.rmcv_pipeline_destroy.argtypes = [C.c_void_p]
    L.rmcv_pipeline_context.restype = C.c_void_p
    L.rmcv_pipeline_context.argtypes = [C.c_void_p, C.c_int]
    L.rmcv_pipeline_set_hot_contexts.restype = C.c_int
    L.rmcv_pipeline_set_hot_contexts.argtypes = [C.c_void_p, C.c_int]
    L.rmcv_pipeline_set_wait_timeout.restype = C.c_int
    L.rmcv_pipeline_set_wait_timeout.argtypes = [C.c_void_p, C.c_int]
    L.rmcv_pixel_ws_launches.restype = C.c_int64
    L.rmcv_pixel_ws_launches.argtypes = []
    L.rmcv_pipeline_context_of.restype = C.c_void_p
    L.rmcv_pipeline_context_of.argtypes = [C.c_void_p, C.c_uint64]
    # the call of the timed region: fixed argument types, so that ctypes converts without looking at the Python objects' types
    L.rmcv_pipeline_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
    L.rmcv_pipeline_submit_legacy.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.rmcv_pipeline_wait.argtypes = [C.c_void_p, C.c_uint64]
    L.rmcv_pipeline_collect.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.rmcv_pipeline_record.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.rmcv_pipeline_gathered.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.rmcv_pipeline_set_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.rmcv_device_free.restype = None
    L.rmcv_device_free.argtypes = [C.c_int, C.c_void_p]
    return L


def lib():
    """load librmcv_hip.so; raises (never falls back) when it has not been built"""
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
    return _lib


def use(L):
    """dev tool (bench.py RMCV_BENCH_AB=lib:...): make another build THE library for the calls that follow; returns the previous one.
    Contexts belong to the build that made them: switch back before touching them."""
    global _lib
    prev, _lib = lib(), L
    return prev


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_pnp_config():
    c = PnpConfig()
    lib().rmcv_default_pnp_config(C.byref(c))
    return c


def default_params(**kw):
    p = Params()
    lib().rmcv_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p
