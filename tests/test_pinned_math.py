"""pinned_math.h (the transcendental functions shared by the CPU oracle and the HIP kernels) against the
host libm the reference itself would link: <= 1 ulp in double, and -- what the path actually consumes --
identical after narrowing to float in all but a vanishing fraction of cases."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <math.h>
#include "rmcv_amd/csrc/pinned_math.h"
void t_sin(const double* x, double* o, int n) { for (int i = 0; i < n; i++) o[i] = pm_sin(x[i]); }
void t_cos(const double* x, double* o, int n) { for (int i = 0; i < n; i++) o[i] = pm_cos(x[i]); }
void t_atan2(const double* y, const double* x, double* o, int n) { for (int i = 0; i < n; i++) o[i] = pm_atan2(y[i], x[i]); }
void t_atan2f(const float* y, const float* x, float* o, int n) { for (int i = 0; i < n; i++) o[i] = pm_atan2f(y[i], x[i]); }
void t_sinf(const float* x, float* o, int n) { for (int i = 0; i < n; i++) o[i] = pm_sinf(x[i]); }
void t_cosf(const float* x, float* o, int n) { for (int i = 0; i < n; i++) o[i] = pm_cosf(x[i]); }
void r_atan2f(const float* y, const float* x, float* o, int n) { for (int i = 0; i < n; i++) o[i] = atan2f(y[i], x[i]); }
void r_sinf(const float* x, float* o, int n) { for (int i = 0; i < n; i++) o[i] = sinf(x[i]); }
void r_cosf(const float* x, float* o, int n) { for (int i = 0; i < n; i++) o[i] = cosf(x[i]); }
double t_fmod180(double x) { return pm_fmod180(x); }
void t_acos(const double* x, double* o, int n) { for (int i = 0; i < n; i++) o[i] = pm_acos(x[i]); }
'''


@pytest.fixture(scope="module")
def pm(tmp_path_factory):
    d = tmp_path_factory.mktemp("pm")
    c = d / "pm.c"
    c.write_text(SRC)
    so = d / "pm.so"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I", ROOT, str(c), "-o", str(so), "-lm"], check=True)
    L = C.CDLL(str(so))
    L.t_fmod180.restype = C.c_double
    L.t_fmod180.argtypes = [C.c_double]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


def test_sin_cos_double(pm):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-7, 7, 400000), np.deg2rad(np.arange(0, 181, dtype=np.float64)),
                        np.float32(rng.uniform(0, 180, 200000)).astype(np.float64) * math.pi / 180.0])
    o = np.empty_like(x)
    for fn, ref in ((pm.t_sin, np.sin), (pm.t_cos, np.cos)):
        fn(_p(x), _p(o), len(x))
        r = ref(x)
        assert ulps(o, r).max() <= 1
        assert np.count_nonzero(o.astype(np.float32) != r.astype(np.float32)) == 0


def test_atan2_double(pm):
    rng = np.random.default_rng(2)
    y = rng.normal(0, 1, 600000) * 10.0 ** rng.integers(-3, 3, 600000)
    x = rng.normal(0, 1, 600000) * 10.0 ** rng.integers(-3, 3, 600000)
    o = np.empty_like(x)
    pm.t_atan2(_p(y), _p(x), _p(o), len(x))
    r = np.arctan2(y, x)
    assert ulps(o, r).max() <= 2
    # what fitEllipseDirect does with it: theta -> degrees -> float
    a = ((math.pi / 2 + 0.5 * o) * 180 / math.pi).astype(np.float32)
    b = ((math.pi / 2 + 0.5 * r) * 180 / math.pi).astype(np.float32)
    assert np.count_nonzero(a != b) <= 2
    assert pm.t_fmod180(180.0) == 0.0 and pm.t_fmod180(269.5) == 89.5 and pm.t_fmod180(12.25) == 12.25


def test_float_variants(pm):
    """pinned float functions are correctly rounded (evaluated in double, rounded once); glibc's are within
    1 ulp of that.  The reference uses them only for `icon` and for accept/reject thresholds."""
    rng = np.random.default_rng(3)
    n = 500000
    y = np.abs(rng.normal(0, 50, n)).astype(np.float32)
    x = np.abs(rng.normal(0, 50, n)).astype(np.float32)
    o, r = np.empty(n, np.float32), np.empty(n, np.float32)
    pm.t_atan2f(_p(y), _p(x), _p(o), n)
    exact = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.count_nonzero(o != exact.astype(np.float32)) <= 1
    pm.r_atan2f(_p(y), _p(x), _p(r), n)
    assert np.abs(o.view(np.int32) - r.view(np.int32)).max() <= 1
    th = rng.uniform(0, math.pi / 2, n).astype(np.float32)
    for mine, theirs, ref in ((pm.t_sinf, pm.r_sinf, np.sin), (pm.t_cosf, pm.r_cosf, np.cos)):
        mine(_p(th), _p(o), n)
        theirs(_p(th), _p(r), n)
        assert np.count_nonzero(o != ref(th.astype(np.float64)).astype(np.float32)) <= 1
        assert np.abs(o.view(np.int32) - r.view(np.int32)).max() <= 1


def test_acos_double(pm):
    rng = np.random.default_rng(9)
    x = np.concatenate([rng.uniform(-1, 1, 200000), np.linspace(-1, 1, 4001), 1 - np.geomspace(1e-16, 1e-3, 2000),
                        -1 + np.geomspace(1e-16, 1e-3, 2000), rng.uniform(-1e-9, 1e-9, 1000)])
    o = np.empty_like(x)
    pm.t_acos(_p(x), _p(o), len(x))
    assert ulps(o, np.arccos(x)).max() <= 1
    assert o[np.argmax(x == 1.0)] == 0.0 if (x == 1.0).any() else True
