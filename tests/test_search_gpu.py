"""Outer searches (fir_ap.m) on the GPU: the reference's bisection order and the batched multi-probe variant."""
import time

import numpy as np
import pytest
from conftest import c13, relinf

import mbfir

pytestmark = pytest.mark.gpu


def test_min_order_search_finds_the_feasibility_anchor():
    """S-C13 is solvable at 58 taps (bSSFP_pulse_sb_mb.m:25); the oracle's threshold is 57 feasible / 56 not."""
    f, a, d = c13(100)
    log1, log4 = [], []
    t0 = time.time()
    h1, s1, n1, f1 = mbfir.fir_ap(100, f, a, d, 1e-3, 1, 0, log=log1)
    t1 = time.time()
    h4, s4, n4, f4 = mbfir.fir_ap(100, f, a, d, 1e-3, 1, 0, probes=4, log=log4)
    t4 = time.time()
    assert s1 == s4 == "Solved" and n1 == n4 == 57 and len(h1) == len(h4) == 57
    assert relinf(h4, h1) <= 1e-9                      # same design problem at the end of both searches
    # the bisection probes ceil((n_top + n_bot) / 2) in the reference's order
    assert [v for k, v, s in log1 if k == "n"][:3] == [51, 76, 64]
    assert len([1 for k, v, s in log4 if k == "n"]) >= len([1 for k, v, s in log1 if k == "n"])
    assert np.allclose(f1, f) and np.allclose(f4, f)
    print("sequential %.2f s, 4 probes per round %.2f s" % (t1 - t0, t4 - t1))


def test_min_transition_search_widens_the_bands():
    f, a, d = c13(80)
    h, s, n_op, f_op = mbfir.fir_ap(80, f, a, d, 1e-3, 0, 1.0, probes=3)
    assert s == "Solved" and n_op == 80 and len(h) == 80
    f = np.asarray(f)
    widen = f[0::2] - f_op[0::2]
    assert np.all(widen > 0) and np.allclose(widen, widen[0]) and np.allclose(f_op[1::2] - f[1::2], widen[0])
    assert widen[0] < np.min(f[2::2] - f[1:-1:2]) / 2
    # the widened spec is still solvable, a little more is not guaranteed to be
    h2, s2 = mbfir.fir_ap_cvx(80, f_op, a, d, 0.1, 1e-3)
    assert s2 == "Solved" and relinf(h2, h) <= 1e-6


def test_argument_errors_and_tight_spec():
    f, a, d = c13(100)
    with pytest.raises(ValueError, match="invalid input of min_tran"):
        mbfir.fir_ap(100, f, a, d, 1e-3, 0, 2)
    with pytest.raises(ValueError, match="invalid input of min_order"):
        mbfir.fir_ap(100, f, a, d, 1e-3, -1, 0)
    with pytest.raises(NotImplementedError):
        mbfir.fir_ap(100, f, a, d, 1e-3, 0, 0, 1)
    with pytest.raises(ValueError, match="original parameters are too tight"):
        mbfir.fir_ap(40, f, a, d, 1e-3, 1, 0)           # infeasible at 40 taps (fir_ap.m:52-54)
