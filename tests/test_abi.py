"""The C-ABI library loads on a GPU-less host and exports every symbol include/rmcv_abi.h declares
(no compute calls here); struct layouts match the header; the product package never touches oracle/."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    h = open(os.path.join(ROOT, "include", "rmcv_abi.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(rmcv_[a-z0-9_]+)\s*\(", h)))


def test_exports_match_header():
    from rmcv_amd import abi
    L = abi.lib()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "librmcv_hip.so does not export %s" % n
    assert sorted(abi.EXPORTS) == names
    assert L.rmcv_abi_version() == 1


def test_struct_layouts_and_defaults():
    from rmcv_amd import abi
    assert C.sizeof(abi.Params) == 56 and C.sizeof(abi.Limits) == 32
    assert abi.POINT.itemsize == 8 and abi.RRECT.itemsize == 20 and abi.LIGHTBLOB.itemsize == 56 and abi.ARMOUR.itemsize == 88
    p = abi.default_params()   # executable/main.cpp:172-176
    assert (p.camp, p.lower_bound, p.morph) == (1, 80, 2)
    assert (p.tilt_max, p.ratio_lo, p.ratio_hi, p.area_lo, p.area_hi) == (70.0, 1.5, 80.0, 10.0, 99999.0)
    assert p.angle_diff_max == 12.0 and p.shear_max == 22.0 and abs(p.length_ratio_max - 0.4) < 1e-7
    lim = abi.Limits()
    abi.lib().rmcv_default_limits(C.byref(lim))
    assert lim.max_frames == 256 and lim.max_points == 65536


def test_no_device_fails_loudly():
    """without a GPU the product refuses to run -- it must never fall back to a CPU path"""
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    from rmcv_amd import Context, RmcvError
    with pytest.raises(RmcvError) as e:
        Context(device=0)
    assert e.value.code == -5


def test_product_does_not_import_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "rmcv_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".hpp")) or f == "Makefile":
                s = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_lib" not in s and "liboracle" not in s, os.path.join(dp, f)
                assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', s), os.path.join(dp, f)
                assert not re.search(r"^\s*(from|import)\s+\S*oracle", s, flags=re.M), os.path.join(dp, f)


def test_synth_is_deterministic():
    from rmcv_amd import synth
    a, b = synth.frame(3, 320, 256), synth.frame(3, 320, 256)
    assert np.array_equal(a, b) and synth.checksum(a) == synth.checksum(b)
    assert not np.array_equal(a, synth.frame(4, 320, 256))
    red = synth.frame(3, 320, 256, camp=0)
    assert np.array_equal(red[..., 0], a[..., 2]) and np.array_equal(red[..., 2], a[..., 0])   # the mirrored stream


def test_gather_entry_points_check_their_arguments():
    """rmcv_comm_* / rmcv_gather (the C-ABI form of the multi-GPU gather): argument checks, no GPU and no RCCL needed"""
    from rmcv_amd import abi
    L = abi.lib()
    h = C.c_void_p()
    idb = (C.c_uint8 * abi.COMM_ID_BYTES)()
    assert L.rmcv_comm_unique_id(None) == abi.ERR_BAD_ARG
    assert L.rmcv_comm_create(None, 2, 0, 0, C.byref(h)) == abi.ERR_BAD_ARG and not h.value
    assert L.rmcv_comm_create(idb, 0, 0, 0, C.byref(h)) == abi.ERR_BAD_ARG
    assert L.rmcv_comm_create(idb, 2, 2, 0, C.byref(h)) == abi.ERR_BAD_ARG        # rank out of range
    assert L.rmcv_comm_create(idb, 2, 0, 0, None) == abi.ERR_BAD_ARG
    import torch
    if torch.cuda.device_count() == 0:
        assert L.rmcv_comm_create(idb, 1, 0, 0, C.byref(h)) == abi.ERR_NO_DEVICE     # like rmcv_ctx_create: no CPU path
    assert L.rmcv_comm_info(None, None, None) == abi.ERR_BAD_ARG
    assert L.rmcv_gather(None, None, C.c_int64(16), None, 0, None) == abi.ERR_BAD_ARG
    L.rmcv_comm_destroy.restype = None
    L.rmcv_comm_destroy(None)                                                        # a no-op


def test_pipeline_entry_points_without_a_gpu():
    """rmcv_pipeline_*: struct layouts, defaults, argument checks; without a GPU creation fails with RMCV_ERR_NO_DEVICE (no CPU path)"""
    import pytest
    import torch
    from rmcv_amd import abi
    L = abi.lib()
    assert C.sizeof(abi.PipelineConfig) == 40 and C.sizeof(abi.PipelineInfo) == 152
    cfg = abi.PipelineConfig()
    L.rmcv_default_pipeline_config(C.byref(cfg))
    assert (cfg.depth, cfg.pixel_streams, cfg.sparse_streams, cfg.sparse_waves, cfg.pixel_groups, cfg.host_results, cfg.dense_streams, cfg.hot_contexts) == (8, 2, 4, 4, 2, 1, 4, 0)
    h = C.c_void_p()
    assert L.rmcv_pipeline_create(0, None, None, None) == abi.ERR_BAD_ARG
    t = C.c_uint64(0)
    assert L.rmcv_pipeline_submit(None, None, 1, 1, 1, 3, 3, None, 15, C.addressof(t)) == abi.ERR_BAD_ARG
    assert L.rmcv_pipeline_collect(None, 0, None, 0, None, None) == abi.ERR_BAD_ARG
    assert L.rmcv_pipeline_drain(None) == abi.ERR_BAD_ARG and L.rmcv_pipeline_wait(None, 0) == abi.ERR_BAD_ARG
    assert L.rmcv_pipeline_context(None, 0) is None
    assert L.rmcv_pipeline_context_of(None, 0) is None and L.rmcv_pipeline_set_hot_contexts(None, 4) == abi.ERR_BAD_ARG
    assert L.rmcv_pixel_ws_launches() == 0
    assert L.rmcv_device_alloc(0, 0, C.byref(h)) == abi.ERR_BAD_ARG
    assert L.rmcv_pipeline_set_wait_timeout(None, 10) == abi.ERR_BAD_ARG
    if not torch.cuda.is_available():
        from rmcv_amd import Pipeline, RmcvError
        with pytest.raises(RmcvError) as e:
            Pipeline(device=0, depth=2)
        assert e.value.code == abi.ERR_NO_DEVICE
        assert L.rmcv_device_alloc(0, 1024, C.byref(h)) == abi.ERR_NO_DEVICE


def test_bench_refuses_silent_dev_knobs():
    """an environment knob that changes the timed region is refused unless --dev (and then echoed in config.dev_knobs)"""
    import sys
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    assert bench.dev_knobs({}, False) == {}
    with pytest.raises(SystemExit) as e:
        bench.dev_knobs({"RMCV_BENCH_STAGES": "1", "HOME": "/x"}, False)
    assert "RMCV_BENCH_STAGES" in str(e.value)
    assert bench.dev_knobs({"RMCV_SPARSE_WAVES": "8", "RMCV_K1_BPC": ""}, True) == {"RMCV_SPARSE_WAVES": "8"}


def test_round5_entry_points_without_a_gpu():
    """the deadline / diagnosis entry points check their arguments; rmcv_hw_queues_hint sets GPU_MAX_HW_QUEUES only where it is unset
    (the library itself no longer touches the environment when it is loaded: ADVICE r4)"""
    import subprocess
    import sys
    from rmcv_amd import abi
    L = abi.lib()
    us = (C.c_double * 9)()
    assert L.rmcv_ctx_frame_timing(None, us, 9) == abi.ERR_BAD_ARG
    assert L.rmcv_pipeline_reset_stats(None) == abi.ERR_BAD_ARG and L.rmcv_pipeline_set_wait_timeout(None, 5) == abi.ERR_BAD_ARG
    assert abi.ERR_TIMEOUT == -7 and abi.OPT_WAIT_TIMEOUT_MS == 15 and abi.OPT_IMAGE_EXPORT == 17
    code = ("import ctypes, os, sys; sys.path.insert(0, %r); "
            "from rmcv_amd import abi; L = abi.lib(); libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; "
            "print(libc.getenv(b'GPU_MAX_HW_QUEUES'), L.rmcv_hw_queues_hint(), libc.getenv(b'GPU_MAX_HW_QUEUES'))" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout.split()
    assert out == ["None", "12", "b'12'"], out                       # loading the library leaves the variable alone; the hint sets it
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, GPU_MAX_HW_QUEUES="6"), capture_output=True, text=True, timeout=120).stdout.split()
    assert out == ["b'6'", "6", "b'6'"], out                         # ... and never overrides the host's own value
