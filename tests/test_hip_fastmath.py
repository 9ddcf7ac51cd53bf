"""Accuracy of the device math helpers (fz_fastmath.h) against NumPy, on the GPU."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(which, x):
    from frankenz_amd.engine import get_engine
    from frankenz_amd._lib import check, ptr
    eng = get_engine()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    check(eng.lib.fz_selftest_math(eng.h, which, ptr(x), len(x), ptr(out)))
    return out


def test_rcp_newton():
    rs = np.random.RandomState(0)
    x = np.exp(rs.uniform(-200, 200, 200000)) * rs.choice([-1, 1], 200000)
    seed = np.abs(run(0, x) * x - 1).max()
    e1 = np.abs(run(1, x) * x - 1).max()
    e2 = np.abs(run(2, x) * x - 1).max()
    print("rcp rel err: seed %.3g, 1 step %.3g, 2 steps %.3g" % (seed, e1, e2))
    assert e1 < 1e-12 and e2 < 5e-16


def test_log_pos():
    rs = np.random.RandomState(1)
    x = np.concatenate([np.exp(rs.uniform(-700, 700, 200000)), rs.uniform(0.5, 2.0, 100000),
                        1 + rs.uniform(-1e-6, 1e-6, 1000), [5e-324, 1e-310, 2.2250738585072014e-308, 1.0]])
    got, want = run(3, x), np.log(x)
    err = np.abs(got - want) / (1 + np.abs(want))
    print("log_pos max scaled err %.3g" % err.max())
    assert err.max() < 1e-15
    sp = run(3, np.array([0.0, np.inf, np.nan, -1.0]))
    assert sp[0] == -np.inf and sp[1] == np.inf and np.isnan(sp[2]) and np.isnan(sp[3])


def test_exp_neg():
    rs = np.random.RandomState(2)
    x = -np.concatenate([rs.uniform(0, 700, 200000), rs.uniform(0, 2, 100000), [0.0, 1e-300, 699.9, 700.0]])
    got, want = run(4, x), np.exp(x)
    rel = np.abs(got / want - 1)
    print("exp_neg max rel err %.3g (|x| < 40: %.3g)" % (rel.max(), rel[np.abs(x) < 40].max()))
    # one-constant argument reduction: the error grows like 3.4e-17 |x|, a third of the
    # rounding error 1.1e-16 |x| the argument itself carries
    assert np.all(rel < 5e-16 + 5e-17 * np.abs(x))
    # below -700 the argument is clamped: a value ~1e-304 instead of an underflow to 0
    tail = run(4, np.array([-700.1, -745.0, -800.0, -1e9, -np.inf, np.nan]))
    assert np.all(tail > 0) and np.all(tail < 1.1e-304)
