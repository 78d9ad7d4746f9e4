"""Tuning knobs of a context (rmcv_ctx_set_option) change the schedule, never the results: every option at every value the header
documents, full path on a small batch, lists compared with the default's byte for byte."""
import numpy as np
import pytest

from rmcv_amd import (CAMP_BLUE, OPT_CONTOUR_TIER, OPT_DENSE_DEFER, OPT_PIXEL_GROUPS, OPT_PIXEL_HALO_NT,
                      OPT_SPARSE_WAVES, STAGE_ALL, Context, default_params, synth)
from rmcv_amd.abi import RmcvError

pytestmark = pytest.mark.gpu


def _lists(c):
    arm, offs = c.armours()
    out = [arm.tobytes(), offs.tobytes(), c.counts()["n_contours"].tobytes(), c.counts()["n_points"].tobytes()]
    for f in (0, c.shape[0] // 2, c.shape[0] - 1):
        pts, co = c.contours(f)
        out += [pts.tobytes(), co.tobytes(), c.blobs(f)[0].tobytes(), c.binary(f).tobytes()]
    return out


def test_every_option_leaves_the_results_alone():
    n = 24
    frames = synth.batch(880, n, 1280, 1024, CAMP_BLUE, 1, threads=16)
    frames[5] = synth.frame(885, 1280, 1024, CAMP_BLUE, 14)                    # one frame beyond the LDS tables
    c = Context(device=0, max_frames=n, max_width=1280, max_height=1024, max_contours=4096)
    c.upload(frames)
    c.run(default_params(), STAGE_ALL)
    c.sync()
    ref = _lists(c)
    cases = [(OPT_SPARSE_WAVES, 4), (OPT_PIXEL_GROUPS, 1), (OPT_PIXEL_GROUPS, 5), (OPT_PIXEL_HALO_NT, 1),
             (OPT_DENSE_DEFER, 1), (OPT_CONTOUR_TIER, 2)]
    for opt, val in cases:
        c.set_option(opt, val)
        if opt == OPT_DENSE_DEFER:
            c.set_option(OPT_SPARSE_WAVES, 4)                                   # (the deferral exists for the 4-wavefront kernel)
        c.run(default_params(), STAGE_ALL)
        c.sync()
        got = _lists(c)
        assert all(a == b for a, b in zip(got, ref)), (opt, val)
    for opt, val in ((6, 1), (8, 0), (9, 3), (10, 1),                           # round 3's hand-over and measurement knobs: removed
                     (OPT_PIXEL_HALO_NT, 2), (OPT_DENSE_DEFER, 2), (99, 0)):
        with pytest.raises(RmcvError):
            c.set_option(opt, val)
    c.close()
