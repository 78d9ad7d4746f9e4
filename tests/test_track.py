"""SURVEY 8f-4, the observable part of the tracker: rm::armour::max_IoU and identity_max (src/core.cpp:124-162).  Host-side
functions of the C-ABI (no device work), checked against the oracle and against hand-computed answers."""
import math

import numpy as np

import oracle_lib as O
from rmcv_amd.api import Context


def arm(x, y, w, h):
    a = np.zeros(1, O.ARMOUR)
    a[0]["bbox"] = (x, y, w, h)
    return a[0]


def test_max_iou_known_answers():
    me = arm(10, 10, 20, 20)
    lst = np.array([arm(100, 100, 5, 5), arm(20, 10, 20, 20), arm(10, 10, 20, 20), arm(12, 12, 4, 4)], O.ARMOUR)
    for f in (O.max_iou, Context.max_iou):
        idx, iou = f(me, lst)
        assert idx == 2 and iou == 1.0
        idx, iou = f(me, lst[:2])
        assert idx == 1 and abs(iou - 200.0 / 600.0) < 1e-7      # half overlap: 200 / (400 + 400 - 200)
        assert f(me, lst[:1]) == (-1, 0.0)
        assert f(me, lst[:0]) == (-1, 0.0)
        assert f(arm(0, 0, 0, 0), lst) [0] == -1                  # an empty box overlaps nothing (0/positive = 0, never > 0)


def test_max_iou_random_against_oracle():
    rng = np.random.default_rng(4)
    for _ in range(200):
        n = int(rng.integers(0, 12))
        lst = np.zeros(n, O.ARMOUR)
        lst["bbox"] = np.stack([rng.uniform(-50, 600, n), rng.uniform(-50, 500, n), rng.uniform(0, 120, n), rng.uniform(0, 120, n)], 1)
        me = arm(*rng.uniform(-50, 500, 2), *rng.uniform(1, 150, 2))
        a, b = O.max_iou(me, lst), Context.max_iou(me, lst)
        assert a[0] == b[0] and np.float32(a[1]).tobytes() == np.float32(b[1]).tobytes()


def test_identity_max():
    for f in (O.identity_max, Context.identity_max):
        assert f({}) == (-1, 0.0)
        mid, p = f({3: 5})
        assert mid == 3 and p == 1.0
        mid, p = f({1: 2, 4: 2, 7: 1})                             # tie: the lower key comes first in std::map order
        assert mid == 1 and abs(p - math.exp(2) / (2 * math.exp(2) + math.exp(1))) < 1e-15
        mid, p = f({5: 1, 2: 9, 8: 3})
        assert mid == 2
    hist = {int(k): int(v) for k, v in zip(range(0, 14, 2), [3, 1, 4, 1, 5, 9, 2])}
    assert O.identity_max(hist) == Context.identity_max(hist)
