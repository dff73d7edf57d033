"""Outer searches (mbfir.search) on a synthetic feasibility predicate: probe sequences of the reference's
bisections (fir_qp.m:56-136, ss/fir_min_order_linprog.m:79-231), hand-traced from the .m control flow.
Host only -- the designer is injected, nothing touches the GPU."""
import numpy as np
import pytest

import mbfir


def fake(thr_odd, thr_even):
    def designer(taps):
        ok = taps >= (thr_odd if taps % 2 else thr_even)
        return (np.ones(taps) if ok else np.zeros(0)), ("Solved" if ok else "Failed")
    return designer


def test_min_order_probe_sequence_odd_wins():
    log = []
    h, st = mbfir.fir_min_order_linprog(40, [0, .2, .3, 1], [1, 1, 0, 0], [.01, .01], log=log, designer=fake(13, 16))
    assert st == "Solved" and len(h) == 13
    assert [t for _, t, _ in log] == [39, 21, 11, 17, 15, 13, 14]      # odd search, then the one even probe below 13


def test_min_order_probe_sequence_even_wins():
    log = []
    h, st = mbfir.fir_min_order_qprog_phs(40, [0, .2, .3, 1], [1, 1, 0, 0], [.01, .01], log=log, designer=fake(13, 12))
    assert st == "Solved" and len(h) == 12
    assert [t for _, t, _ in log] == [39, 21, 11, 17, 15, 13, 14, 8, 12, 10]


def test_min_order_parity_restrictions_and_failure():
    h, st = mbfir.fir_min_order_linprog(40, [0, 1], [1, 1], [.01], 1, designer=fake(13, 12))
    assert len(h) == 13                                                # odd only
    h, st = mbfir.fir_min_order_linprog(40, [0, 1], [1, 1], [.01], 2, designer=fake(13, 12))
    assert len(h) == 12                                                # even only
    h, st = mbfir.fir_min_order_linprog(40, [0, 1], [1, 1], [.01], 7, designer=fake(13, 12))
    assert len(h) == 12                                                # anything else: both
    log = []
    h, st = mbfir.fir_min_order_linprog(40, [0, 1], [1, 1], [.01], log=log, designer=fake(99, 99))
    assert st == "Failed" and len(h) == 0
    assert [t for _, t, _ in log] == [39, 40]                          # the longest filters fail: both loops end at once
    with pytest.raises(ValueError):
        mbfir.fir_min_order_linprog(40, [0, 1], [1, 1], None)


def test_min_order_tie_goes_to_even_and_tiny_n():
    # equal lengths cannot happen; "not shorter" odd loses (:226-229): odd 15 vs even 14 -> even
    h, st = mbfir.fir_min_order_linprog(31, [0, 1], [1, 1], [.01], designer=fake(15, 14))
    assert len(h) == 14
    log = []
    h, st = mbfir.fir_min_order_linprog(3, [0, 1], [1, 1], [.01], log=log, designer=fake(1, 2))
    assert st == "Failed" and log == []                                # n_top - n_bot = 1 from the start: no probe at all (:79)
    h, st = mbfir.fir_min_order_linprog(5, [0, 1], [1, 1], [.01], 1, log=log, designer=fake(1, 2))
    assert [t for _, t, _ in log] == [5, 3] and len(h) == 3            # never probes 1 tap (n_bot = 1 is taken as failed)


@pytest.mark.parametrize("probes", [2, 3, 4])
def test_min_order_speculative_probes_reach_the_same_answer(probes):
    for thr_odd, thr_even in ((13, 16), (13, 12), (27, 30), (3, 2), (39, 40)):
        ref = mbfir.fir_min_order_linprog(40, [0, 1], [1, 1], [.01], designer=fake(thr_odd, thr_even))
        log = []
        got = mbfir.fir_min_order_linprog(40, [0, 1], [1, 1], [.01], probes=probes, log=log, designer=fake(thr_odd, thr_even))
        assert got[1] == ref[1] and len(got[0]) == len(ref[0])


def qp_designer(df_min, n_min):
    def designer(n, f):
        ok = (f[2] - f[1]) / 2 >= df_min - 1e-15 and n >= n_min
        return (np.ones(n) if ok else np.zeros(0)), ("Solved" if ok else "Failed")
    return designer


def test_fir_qp_transition_bisection_sequence():
    log = []
    f = [-0.2, 0.2, 0.4, 1.0]
    h, st = mbfir.fir_qp(64, f, [1, 1, 0, 0], [.01, .01], 0, 1, log=log, designer=qp_designer(0.03, 30))
    dfs = [v for k, v, _ in log if k == "df"]
    assert np.allclose(dfs, [0.05, 0.025, 0.0375, 0.03125, 0.028125, 0.0296875, 0.03046875])
    assert log[-1][0] == "df_final" and abs(log[-1][1] - 0.03046875) < 1e-15 and st == "Solved" and len(h) == 64


def test_fir_qp_order_bisection_and_fraction():
    log = []
    f = [-0.2, 0.2, 0.4, 1.0]
    h, st = mbfir.fir_qp(64, f, [1, 1, 0, 0], [.01, .01], 1, 0, log=log, designer=qp_designer(0.03, 30))
    assert [v for k, v, _ in log if k == "n"] == [33, 18, 26, 30, 28, 29] and len(h) == 30
    log = []
    h, st = mbfir.fir_qp(64, f, [1, 1, 0, 0], [.01, .01], 0.5, 0, log=log, designer=qp_designer(0.03, 30))
    assert log[-1] == ("n_final", 47, "Solved") and len(h) == 47        # ceil(64*0.5 + 30*0.5)


def test_fir_qp_argument_errors():
    f = [-0.2, 0.2, 0.4, 1.0]
    with pytest.raises(ValueError, match="too tight"):
        mbfir.fir_qp(20, f, [1, 1, 0, 0], [.01, .01], designer=qp_designer(0.03, 30))
    with pytest.raises(ValueError, match="min_tran"):
        mbfir.fir_qp(64, f, [1, 1, 0, 0], [.01, .01], 0, 1.5, designer=qp_designer(0.03, 30))
    with pytest.raises(ValueError, match="min_order"):
        mbfir.fir_qp(64, f, [1, 1, 0, 0], [.01, .01], 1.5, 0, designer=qp_designer(0.03, 30))
    with pytest.raises(ValueError, match="not enough"):
        mbfir.fir_qp(64, f, [1, 1, 0, 0], None)
