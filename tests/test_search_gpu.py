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
    h0, s0, _, _ = mbfir.fir_ap(100, f, a, d, 1e-3, 0, 0, 1)   # no search: early return, min_peak not applied (fir_ap.m:55-59)
    assert s0 == "Solved" and len(h0) == 100
    with pytest.raises(ValueError, match="original parameters are too tight"):
        mbfir.fir_ap(40, f, a, d, 1e-3, 1, 0)           # infeasible at 40 taps (fir_ap.m:52-54)


def _oracle_designer(name, f, a, d):
    from oracle import designers
    return lambda taps: getattr(designers, name)(taps, f, a, d)


@pytest.mark.parametrize("which,spec", [
    ("fir_min_order_linprog", ([0, 0.2, 0.3, 1], [1, 1, 0, 0], [0.01, 0.01])),
    ("fir_min_order_linprog", ([-1, -0.4, -0.2, 0.3, 0.5, 1], [0, 0, 1, 0.8, 0, 0], [0.01, 0.02, 0.01])),
    ("fir_min_order_qprog_phs", ([-0.6, -0.3, -0.1, 0.1, 0.3, 0.6], [0, 0, 1, 1, 0, 0], [0.02, 0.05 * np.exp(0.3j), 0.02])),
])
def test_min_order_searches_follow_the_oracle_probe_for_probe(which, spec):
    """ss/fir_min_order_*.m on the device against the same search driven by the CPU oracle: same probe
    sequence, same verdict at every probe, same final taps."""
    f, a, d = spec
    base = "fir_linprog" if "linprog" in which else "fir_qprog_phs"
    lg, lo = [], []
    hg, sg = getattr(mbfir, which)(48, f, a, d, log=lg)
    ho, so = getattr(mbfir, which)(48, f, a, d, log=lo, designer=_oracle_designer(base, f, a, d))
    assert lg == lo and sg == so == "Solved"
    assert len(hg) == len(ho) and relinf(hg, ho) <= 1e-6
    # one tap fewer of the same parity is infeasible (that is what the bisection certifies)
    if len(hg) > 2:
        assert getattr(mbfir, base)(len(hg) - 2, f, a, d)[1] == "Failed"
    h4, s4 = getattr(mbfir, which)(48, f, a, d, probes=4)
    assert s4 == "Solved" and len(h4) == len(hg) and relinf(h4, hg) <= 1e-9


def test_min_order_search_reports_failure_when_the_longest_filter_fails():
    h, s = mbfir.fir_min_order_linprog(12, [0, 0.2, 0.22, 1], [1, 1, 0, 0], [0.001, 0.001])
    assert s == "Failed" and len(h) == 0


def test_fir_qp_search_matches_the_oracle():
    """fir_qp.m (which designs through fir_ap_cvx with lambda = 1e5): order bisection of a low-pass spec."""
    from oracle import designers
    f, a, d = [-0.25, 0.25, 0.45, 1.0], [0.15, 0.15, 0, 0], [0.004, 0.002]   # default Peak=1e-3 caps the energy at n/1000
    lg, lo = [], []
    hg, sg = mbfir.fir_qp(40, f, a, d, 1, 0, log=lg)
    ho, so = mbfir.fir_qp(40, f, a, d, 1, 0, log=lo, designer=lambda n, ff: designers.fir_ap_cvx(n, ff, a, d, 1e5))
    assert lg == lo and sg == so == "Solved" and len(hg) == len(ho)
    assert relinf(hg, ho) <= 1e-6
    # transition bisection at 20 taps (at 40 taps with lambda = 1e5 every probe near the threshold ends at the
    # reduced-accuracy wall in both solvers and the verdicts there are a coin toss -- tools/gpu_qp_search.py)
    lg, lo = [], []
    hg, sg = mbfir.fir_qp(20, f, a, d, 0, 0.5, log=lg)
    ho, so = mbfir.fir_qp(20, f, a, d, 0, 0.5, log=lo, designer=lambda n, ff: designers.fir_ap_cvx(n, ff, a, d, 1e5))
    assert lg == lo and sg == so == "Solved" and len(hg) == 20 and lg[-1][0] == "df_final" and lg[-1][1] < 0.1
    assert relinf(hg, ho) <= 1e-6


def test_numerical_failure_is_retried_in_extended_precision_on_the_same_path():
    """fir_ap_cvx(20, ..., 1e5) (fir_qp.m's lambda): the double-precision normal equations hit the numerical wall at
    iteration 19 on the lattice path (cond(H) ~ 1e14; the dense Gram products last a few iterations longer).  The entry
    point retries by itself with the extended-precision KKT solve -- on the SAME lattice path -- and follows the oracle to
    a clean full-accuracy solve in the oracle's 25 iterations; forcing ddkkt from the start gives the same without the
    first attempt."""
    from oracle import designers
    f, a, d = [-0.25, 0.25, 0.45, 1.0], [0.15, 0.15, 0, 0], [0.004, 0.002]
    h, s, i = mbfir.fir_ap_cvx(20, f, a, d, 1e5, info=True)
    ho, so, io = designers.fir_ap_cvx(20, f, a, d, 1e5, info=True)
    assert s == so == "Solved" and i["lattice"] == 1 and i["dd_iters"] > 0 and i["dres"] <= 1e-8
    assert i["relgap"] <= 1e-8 or i["gap"] <= 1e-10           # the solver's own stopping rule (either measure)
    assert relinf(h, ho) <= 1e-6
    h2, s2, i2 = mbfir.fir_ap_cvx(20, f, a, d, 1e5, info=True, opts=mbfir.make_opts(ddkkt=1))
    # (the same path as the oracle's up to the end game's last iterations: rounding decides whether its target is met one iteration
    #  earlier or later)
    assert s2 == "Solved" and i2["lattice"] == 1 and abs(i2["iters"] - io["iters"]) <= 4 and relinf(h2, ho) <= 1e-6
    h3, s3, i3 = mbfir.fir_ap_cvx(20, f, a, d, 1e5, info=True, opts=mbfir.make_opts(ddkkt=-1))     # extended precision forbidden:
    assert s3 == "Solved" and i3["lattice"] == 0 and relinf(h3, ho) <= 1e-6                      # the dense path is the last resort


def test_retry_inside_a_batch():
    """The retry runs per job inside mbfir_solve_batch's worker threads (own context, own arena)."""
    from oracle import designers
    f, a, d = [-0.25, 0.25, 0.45, 1.0], [0.15, 0.15, 0, 0], [0.004, 0.002]
    jobs = [("fir_ap_cvx", (20, f, a, d, 1e5, 1e-3)), ("fir_ap_cvx", (24, f, a, d, 0.1, 1e-2)), ("fir_ap_cvx", (20, f, a, d, 1e5, 1e-3)),
            ("fir_linprog", (33, [0, 0.25, 0.45, 1], [1, 1, 0, 0], [0.02, 0.02])), ("fir_ap_cvx", (20, f, a, d, 1e5, 1e-3))]
    res = mbfir.solve_batch(jobs, streams=4, info=True)
    ho, so = designers.fir_ap_cvx(20, f, a, d, 1e5, 1e-3)
    for k in (0, 2, 4):
        h, s, i = res[k]
        assert s == so == "Solved" and i["lattice"] == 1 and i["dd_iters"] > 0 and relinf(h, ho) <= 1e-6
    assert res[1][1] == "Solved" and res[1][2]["lattice"] == 1 and res[3][1] == "Solved"


def test_extended_precision_solve_is_reproducible_run_to_run():
    """The 57-tap threshold design of S-C13 with the extended-precision KKT solve switched on: the strong rows are put
    in a fixed order on the device (k_dd_order), so two runs give the same bits (the atomic slot counter alone left 1e-7
    between runs of this ill-conditioned design)."""
    f, a, d = c13(100)
    o = mbfir.make_opts(ddkkt=1)
    h1, s1, i1 = mbfir.fir_ap_cvx(57, f, a, d, 0.1, 1e-3, info=True, opts=o)
    h2, s2, i2 = mbfir.fir_ap_cvx(57, f, a, d, 0.1, 1e-3, info=True, opts=o)
    assert s1 == s2 == "Solved" and i1["dd_iters"] > 0 and i1["dd_iters"] == i2["dd_iters"]
    assert np.array_equal(h1, h2) and i1["pcost"] == i2["pcost"]
