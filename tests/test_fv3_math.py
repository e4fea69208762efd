"""include/fv3_math.h: the deterministic exp / log shared by the HIP column kernels and the oracle.
Accuracy against 60-digit decimal arithmetic (< 1 ulp) and the IEEE special cases."""
import math
from decimal import Decimal, getcontext

import numpy as np

import oracle_lib as O


def _worst_ulp(xs, ys, exact):
    getcontext().prec = 60
    worst = Decimal(0)
    for x, y in zip(xs, ys):
        worst = max(worst, abs(Decimal(float(y)) - exact(Decimal(float(x)))) / Decimal(math.ulp(float(y))))
    return float(worst)


def test_exp_below_one_ulp():
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-20, 20, 20000), rng.uniform(-700, 700, 5000), rng.uniform(-1, 1, 5000) * 1e-3,
                         rng.uniform(-0.4, 0.4, 10000), 0.2857 * np.log(rng.uniform(1.0, 1.1e5, 20000))])
    assert _worst_ulp(xs, O.fexp(xs), lambda d: d.exp()) < 1.0


def test_log_below_one_ulp():
    rng = np.random.default_rng(2)
    xs = np.concatenate([np.exp(rng.uniform(-30, 30, 20000)), rng.uniform(0.5, 2.0, 20000), 1 + rng.uniform(-1, 1, 5000) * 1e-4,
                         np.exp(rng.uniform(-700, 700, 3000)), rng.uniform(1.0, 1.1e5, 20000)])
    assert _worst_ulp(xs, O.flog(xs), lambda d: d.ln()) < 1.0


def test_special_values():
    with np.errstate(all="ignore"):
        x = np.array([0.0, -0.0, -800.0, 800.0, np.inf, -np.inf, np.nan, 709.78, -745.0, -708.0, 1.0])
        got, ref = O.fexp(x), np.exp(x)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        m = ~np.isnan(ref)
        assert np.array_equal(np.isinf(got[m]), np.isinf(ref[m]))
        f = m & np.isfinite(ref) & (np.abs(ref) > 1e-300)
        assert np.allclose(got[f], ref[f], rtol=3e-16, atol=0.0) and got[2] == 0.0 and got[0] == 1.0
        x = np.array([0.0, -0.0, -1.0, np.inf, np.nan, 5e-324, 1e-310, 1.0, 2.0])
        got, ref = O.flog(x), np.log(x)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        m = ~np.isnan(ref)
        assert np.array_equal(got[m] == -np.inf, ref[m] == -np.inf) and got[3] == np.inf and got[7] == 0.0
        f = m & np.isfinite(ref)
        assert np.allclose(got[f], ref[f], rtol=3e-16, atol=0.0)
