"""The C++ / OpenMP twin of the oracle's solver (oracle/cpu_ipm.cpp, bench.py's cpu_baseline) against the NumPy oracle:
same verdicts, same iteration counts, solutions equal to rounding, independent of the thread count."""
import numpy as np
import pytest

from oracle import assemble, conic_ipm, cpu_ipm

F6 = [-0.6, -0.35, -0.1, 0.15, 0.45, 0.8]
A6 = [0, 0, 0.7, 0.7, 0, 0]
D3 = [0.01, 0.02, 0.01]

CASES = {
    "ap_feasible": lambda: assemble.assemble_fir_ap_cvx(33, F6, A6, D3, 0.1, 1e-2, 0),
    "ap_infeasible": lambda: assemble.assemble_fir_ap_cvx(8, F6, A6, D3, 0.1, 1e-2, 0),
    "ap_odd_columns": lambda: assemble.assemble_fir_ap_cvx(21, F6, A6, [0.03, 0.05, 0.03], 0.05, 1e-1, 0),
    "qp_feasible": lambda: assemble.assemble_fir_qp_cvx(32, [0, 0.2, 0.4, 1], [1, 1, 0, 0], [0.1, 0.1], 3.0, 1.0, 0),
    "qp_infeasible": lambda: assemble.assemble_fir_qp_cvx(32, [0, 0.2, 0.4, 1], [1, 1, 0, 0], [0.1, 0.1], 120.0, 0.0, 0),
    "linprog": lambda: assemble.assemble_fir_linprog(41, [0, 0.2, 0.35, 1], [1, 1, 0, 0], [0.02, 0.02], 0),
    "qprog_phs": lambda: assemble.assemble_fir_qprog_phs(23, [-1, -0.6, -0.2, 0.2], [1, 1, 0, 0], [0.05 * np.exp(0.3j), 0.02], 0),
}


def _both(P, **kw):
    args = (P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"])
    return conic_ipm.solve(*args, **kw), cpu_ipm.solve(*args, **kw)


@pytest.mark.parametrize("name", sorted(CASES))
def test_cpp_solver_matches_numpy_oracle(name):
    r, q = _both(CASES[name]())
    assert q["status"] == r["status"], (name, q["status"], r["status"])
    assert abs(q["iters"] - r["iters"]) <= 1, (q["iters"], r["iters"])
    if r["status"] == conic_ipm.STATUS_OPTIMAL:
        assert abs(q["pcost"] - r["pcost"]) <= 1e-9 * max(1.0, abs(r["pcost"]))
        assert np.abs(q["x"] - r["x"]).max() <= 1e-7 * np.abs(r["x"]).max()
        assert q["pres"] <= 1e-8 and q["dres"] <= 1e-8


def test_at_least_one_case_of_each_verdict():
    st = {n: _both(CASES[n]())[0]["status"] for n in ("ap_feasible", "ap_infeasible")}
    assert st["ap_feasible"] == conic_ipm.STATUS_OPTIMAL and st["ap_infeasible"] == conic_ipm.STATUS_PRIMAL_INFEASIBLE


def test_thread_count_does_not_change_the_result():
    P = CASES["ap_feasible"]()
    args = (P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"])
    a = cpu_ipm.solve(*args, threads=1)
    b = cpu_ipm.solve(*args, threads=4)
    assert a["threads"] == 1 and b["threads"] == 4
    assert a["status"] == b["status"] == 0 and a["iters"] == b["iters"]
    assert np.abs(a["x"] - b["x"]).max() <= 1e-9 * np.abs(a["x"]).max()


def test_iteration_cap_and_bad_cone():
    P = CASES["ap_feasible"]()
    r = cpu_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"], max_iter=3)
    assert r["status"] == conic_ipm.STATUS_MAXIT and r["iters"] == 3
    with pytest.raises(ValueError):
        cpu_ipm.solve(P["c"], P["G"], P["h"], P["l"] + 1, P["nq3"], P["big"])
