"""rmcv_amd -- MI355X-native (gfx950, hand-written HIP) implementation of rmcv's per-frame
armour-detection hot path, behind the reference's own function names.

    from rmcv_amd import Context
    ctx = Context(device=0)
    contours, binary = ctx.extract_color(image, CAMP_BLUE, 80)
"""
from .abi import (ARMOUR, CAMP_BLUE, CAMP_GUIDELIGHT, CAMP_NEUTRAL, CAMP_RED, LIGHTBLOB, MORPH_CLOSE, MORPH_DILATE,
                  MORPH_NONE, FRAME_MID_PATH, FRAME_SLOW_PATH, OPT_CONTOUR_TIER, OPT_FRAME_UPLOAD, OPT_DENSE_DEFER, OPT_PIXEL_HALO_NT, OPT_OVERLOADS, OPT_PIXEL_GROUPS, OPT_PIXEL_SHAPE, OPT_WAIT_TIMEOUT_MS, OPT_TEST_DELAY_US, OPT_IMAGE_EXPORT, OPT_TEST_SLOW_US, OPT_RUN_AHEAD, OPT_SPARSE_WAVES, POINT, RRECT, STAGE_ALL, STAGE_ARMOURS, STAGE_BINARY, STAGE_BLOBS, STAGE_CONTOURS, STAGE_IDENTITY, STAGE_NO_IMAGE, STAGE_POSE,
                  SVM_FEATURES, LegacyParams, Limits,
                  Params, PnpConfig, RmcvError, default_params, default_pnp_config)
from .api import Context
from .pipeline import Pipeline

__all__ = [n for n in dir() if not n.startswith("_")]
