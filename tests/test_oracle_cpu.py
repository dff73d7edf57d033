"""CPU tests of the oracle (no GPU): golden vectors, the independent HiGHS cross-check,
structural identities and the reference's return / error conventions."""
import warnings

import numpy as np
import pytest
from conftest import CASES, c13, relinf

from oracle import assemble, conic_ipm, designers, specfact

warnings.filterwarnings("ignore", category=RuntimeWarning)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name, golden):
    fn, args = CASES[name]
    h, status, info = getattr(designers, fn)(*args, info=True)
    g = golden[name]
    assert status == g["status"]
    if status == "Solved":
        assert abs(info["pcost"] - g["pcost"]) <= 1e-9 * max(1.0, abs(g["pcost"]))
        hg = np.array(g["h"]["re"]) + 1j * np.array(g["h"]["im"])
        assert relinf(h, hg) <= 1e-7
        assert len(h) == args[0]
    else:
        assert len(h) == 0                      # h = [] on failure


@pytest.mark.parametrize("name", [k for k in sorted(CASES) if CASES[k][0] in ("fir_linprog", "fir_ap_cvx")])
def test_golden_objective_matches_highs(name, golden):
    """Independent solver pin: the oracle's optimum equals the HiGHS optimum on every pure-LP
    instance (fir_linprog, and fir_ap_cvx with slack spike cones)."""
    g = golden[name]
    if "highs_obj" not in g:
        pytest.skip("not a pure LP (cones active or infeasible)")
    assert abs(g["pcost"] - g["highs_obj"]) <= 1e-7 * max(1.0, abs(g["highs_obj"]))
    assert g["highs_x_maxdiff"] <= 1e-7        # unique optimiser: HiGHS vertex == IPM limit point


@pytest.mark.parametrize("name", ["ap_c13_58", "qp_modelA48", "qp_modelB25", "qphs21", "qphs22"])
def test_golden_cone_programs_are_pinned_by_an_unrelated_method(name, golden):
    """The programs with ACTIVE cones, which no LP solver covers (tests/golden/make_golden.py (3)): Kelley's cutting
    planes on HiGHS for the spike cones of fir_ap_cvx, Lawson-Hanson least-distance programming on NNLS for
    fir_qprog_phs, SLSQP from a cutting-plane start for fir_qp_cvx -- none shares a line with the interior-point
    iteration of oracle / C++ twin / device.  The pin's own point is feasible to 1e-10 and its objective equals the
    oracle's; the optimisers agree as far as the flatness of each objective lets them."""
    g = golden[name]
    pin = g["pin"]
    assert pin["viol"] <= 1e-10                                           # the pin's point is feasible
    assert abs(pin["obj"] - g["pcost"]) <= 2e-9 * max(1.0, abs(g["pcost"]))
    assert pin["x_maxdiff"] <= {"fir_ap_cvx": 1e-9, "fir_qprog_phs": 1e-7, "fir_qp_cvx": 2e-6}[g["designer"]]
    if "lower_bound" in pin and pin["lower_bound"] is not None:           # every cutting-plane LP bounds the optimum from below
        assert pin["lower_bound"] <= g["pcost"] + 1e-9 * max(1.0, abs(g["pcost"]))
    if name == "ap_c13_58":                                               # converged cutting planes: a two-sided bracket
        assert g["pcost"] - pin["lower_bound"] <= 1e-10


def test_live_cutting_plane_pin_of_active_spike_cones():
    """The same cutting-plane loop run live on a small fir_ap_cvx program whose spike cones are active."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    from conftest import A_C13, D_C13, F100
    args = (30, [2.0 * x for x in F100], A_C13, D_C13, 0.1, 2e-3)         # the C-13 spec, bands twice as wide, 30 taps
    P = assemble.assemble_fir_ap_cvx(*args)
    r = conic_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"])
    assert r["status"] == conic_ipm.STATUS_OPTIMAL
    cp = make_golden.cutting_plane(P)
    assert cp["viol"] <= 1e-10
    q = (P["h"][P["l"]:] - P["G"][P["l"]:] @ r["x"]).reshape(-1, 3)
    assert (q[:, 0] - np.hypot(q[:, 1], q[:, 2])).min() <= 1e-7 * q[:, 0].max()        # some spike cone IS active
    assert abs(cp["obj"] - r["pcost"]) <= 1e-9 * max(1.0, abs(r["pcost"]))
    assert np.abs(cp["x"] - r["x"]).max() <= 1e-8


def test_headline_fixture_is_certified_and_pinned_by_highs():
    """tests/golden/c3_golden.json (the BASELINE headline instance: n=512, 16384 grid points, fir_ap_cvx form): the
    oracle's point carries a primal-dual certificate re-evaluated in plain NumPy, and HiGHS -- on the LP relaxation,
    whose optimum leaves every spike cone slack and therefore IS the SOCP optimum -- returns the same objective within
    the two gaps and the same x to 2e-10."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c3_golden.json")) as fh:
        g = json.load(fh)["c3_ap_512_16384"]
    c = g["certificate"]
    assert g["status"] == 0 and g["chol_fixes"] == 0
    assert c["pres"] <= 1e-10 and c["dres"] <= 1e-9 and c["s_outside"] >= -1e-12 and c["z_outside"] >= -1e-12
    assert 0 <= c["pcost"] - c["dcost"] <= 1e-9 * max(1.0, abs(c["pcost"])) and abs(c["sz"] - (c["pcost"] - c["dcost"])) <= 1e-12
    assert g["highs_status"] == 0 and g["highs_cones_slack"] and g["min_spike_cone_slack"] > 0
    assert abs(g["highs_obj"] - g["pcost"]) <= 1e-10                       # (objective ~6.5e-4: 3e-8 relative, inside relgap 9e-8)
    # HiGHS's optimum against the oracle's own bracket [dcost, pcost]: since the end game of round 6 the bracket is 1e-11 wide --
    # narrower than what HiGHS's 1e-10 feasibility tolerances leave of its objective
    assert c["dcost"] - 1e-10 <= g["highs_obj"] <= c["pcost"] + 1e-10
    assert g["highs_x_maxdiff"] <= 1e-9
    assert len(g["x"]) == 1024 and len(g["h_re"]) == 512


def test_live_highs_crosscheck():
    from scipy.optimize import linprog
    P = assemble.assemble_fir_linprog(33, [0, 0.25, 0.45, 1], [1, 1, 0, 0], [0.02, 0.02])
    r = conic_ipm.solve(P["c"], P["G"], P["h"], P["l"])
    rh = linprog(P["c"], A_ub=P["G"], b_ub=P["h"], bounds=[(None, None)] * len(P["c"]), method="highs")
    assert r["status"] == conic_ipm.STATUS_OPTIMAL and rh.status == 0
    assert abs(r["pcost"] - rh.fun) <= 1e-6 * abs(rh.fun)
    assert (P["G"] @ r["x"] - P["h"]).max() <= 1e-9        # primal feasible


def test_kkt_certificate_of_a_socp_solution():
    fn, args = CASES["ap_c13_58"]
    P = assemble.assemble_fir_ap_cvx(*args)
    r = conic_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"])
    assert r["status"] == conic_ipm.STATUS_OPTIMAL
    x, s, z = r["x"], r["s"], r["z"]
    l = P["l"]
    assert np.linalg.norm(P["G"] @ x + s - P["h"]) <= 1e-9
    assert np.linalg.norm(P["G"].T @ z + P["c"]) <= 1e-7
    assert s[:l].min() > 0 and z[:l].min() > 0
    sq, zq = s[l:].reshape(-1, 3), z[l:].reshape(-1, 3)
    assert (sq[:, 0] - np.hypot(sq[:, 1], sq[:, 2])).min() > -1e-12
    assert (zq[:, 0] - np.hypot(zq[:, 1], zq[:, 2])).min() > -1e-12
    assert abs(s @ z) <= 1e-8
    # the spike cones are ACTIVE here (objective above the LP relaxation's 0.0115029)
    assert (sq[:, 0] - np.hypot(sq[:, 1], sq[:, 2])).min() < 1e-6
    assert r["pcost"] > 0.01150289 + 1e-4


def test_feasibility_anchor_min_order_58():
    """bSSFP_pulse_sb_mb.m:25 -- the C-13 lactate spec is solvable at 58 taps; below it fails."""
    f, a, d = c13(58)
    assert designers.fir_ap_cvx(58, f, a, d, 0.1, 1e-3)[1] == "Solved"
    assert designers.fir_ap_cvx(50, f, a, d, 0.1, 1e-3)[1] == "Failed"
    assert designers.fir_ap_cvx(30, f, a, d, 0.1, 1e-3)[1] == "Failed"


def test_gram_is_toeplitz_plus_hankel():
    """sum_i d_i cos(j w_i) cos(k w_i) = (C(j-k) + C(j+k)) / 2 with C(l)=sum d_i cos(l w_i)."""
    n = 12
    P = assemble.assemble_fir_ap_cvx(n, [-0.5, -0.2, 0.1, 0.4], [0, 0, 1, 1], [0.01, 0.02], 0.1, 1e-2)
    A, w = P["meta"]["A"], P["meta"]["w"]
    d = np.random.default_rng(1).random(len(w)) + 0.1
    Gm = (A.T * d) @ A
    C = lambda l: np.sum(d * np.cos(l * w))
    S = lambda l: np.sum(d * np.sin(l * w))
    for j in range(1, n):
        for k in range(1, n):
            assert abs(Gm[j, k] - 2 * (C(j - k) + C(j + k))) <= 1e-9 * abs(Gm).max()            # 2cos * 2cos
            assert abs(Gm[n - 1 + j, n - 1 + k] - 2 * (C(j - k) - C(j + k))) <= 1e-9 * abs(Gm).max()
            assert abs(Gm[j, n - 1 + k] - 2 * (S(j + k) - S(j - k))) <= 1e-9 * abs(Gm).max()


def test_spectral_factor_reproduces_a_positive_spectrum():
    """fir_ap_cvx.m:208-217: |fft(h)|^2 == fft(r) when the spectrum is strictly positive."""
    rng = np.random.default_rng(3)
    n = 24
    h0 = 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    h0[0] = 1.0                                        # zeros well inside the unit circle
    r2 = np.correlate(h0, h0, mode="full")[n - 1:]
    x = np.concatenate([[r2[0].real], r2[1:].real, r2[1:].imag])
    h = specfact.fmp2(specfact.x_to_r(x, n))
    L = 4096
    S = np.abs(np.fft.fft(h0, L)) ** 2
    assert relinf(np.abs(np.fft.fft(h, L)) ** 2, S) <= 1e-9
    roots = np.roots(h)
    assert np.abs(roots).max() <= 1 + 1e-6            # minimum phase: all zeros inside the unit circle
    assert h.shape == (n,)


@pytest.mark.parametrize("name", ["lin_real64", "lin_real33", "lin_cplx31", "lin_cplx32"])
def test_linprog_response_equals_dft_of_taps(name, golden):
    """ss/fir_linprog.m:254-263: A*x is the (zero-phase) frequency response of fill_h(x)."""
    fn, args = CASES[name]
    P = assemble.assemble_fir_linprog(*args)
    x = np.array(golden[name]["x"])
    h = np.array(golden[name]["h"]["re"]) + 1j * np.array(golden[name]["h"]["im"])
    n = args[0]
    w = P["meta"]["w"]
    k = np.arange(n) - (n - 1) / 2.0
    H = np.exp(-1j * np.outer(w, k)) @ h               # sum_k h_k e^{-j w (k - (n-1)/2)}: zero-phase response
    assert np.abs(H.imag).max() <= 1e-9
    assert relinf(H.real, P["meta"]["A"] @ x) <= 1e-9
    U, L = P["meta"]["U_b"], P["meta"]["L_b"]
    assert (H.real - U).max() <= 1e-7 and (L - H.real).max() <= 1e-7


def test_qp_solution_meets_its_cones(golden):
    fn, args = CASES["qp_modelA48"]
    P = assemble.assemble_fir_qp_cvx(*args)
    x = np.array(golden["qp_modelA48"]["x"])
    n = args[0]
    s = P["h"] - P["G"] @ x
    q = s[: 3 * P["nq3"]].reshape(-1, 3)
    assert (q[:, 0] - np.hypot(q[:, 1], q[:, 2])).min() >= -1e-8
    sb = s[3 * P["nq3"]:]
    assert sb[0] - np.linalg.norm(sb[1:]) >= -1e-8
    h = x[:n] + 1j * x[n:2 * n]
    assert abs(x[2 * n] - np.linalg.norm(x[:2 * n])) <= 1e-6          # E_total = ||x|| at the optimum
    assert abs(x[2 * n + 1] - np.abs(h).max()) <= 1e-6                # Peak = max |h_i|


def test_error_and_early_fail_conventions():
    with pytest.raises(ValueError, match="invalid input of obj"):
        assemble.assemble_fir_ap_cvx(10, [0, 0.2, 0.4, 1], [1, 1, 0, 0], [0.1, 0.1], obj=-1.0)
    with pytest.raises(ValueError, match="invalid input of obj"):
        assemble.assemble_fir_qp_cvx(10, [0, 0.2, 0.4, 1], [1, 1, 0, 0], [0.1, 0.1], 1.0, [1, 2, 3])
    with pytest.raises(ValueError, match="sloped"):
        assemble.assemble_fir_qprog_phs(11, [0, 0.2, 0.4, 1], [1, 0.5, 0, 0], [0.1, 0.1])
    with pytest.raises(ValueError, match="straddling"):
        assemble.assemble_fir_qprog_phs(11, [0, 0.2, 0.4, 1], [0.05, 0.05, 0, 0], [0.1 * np.exp(0.2j), 0.1])
    # even length with a==1 at Nyquist: Failed before any solve (ss/fir_linprog.m:66-75)
    h, status = designers.fir_linprog(32, [0, 0.2, 0.3, 1], [0, 0, 1, 1], [0.01, 0.01])
    assert status == "Failed" and len(h) == 0
    h, status = designers.fir_qprog_phs(22, [-1, -0.6, -0.2, 0.2], [1, 1, 0, 0], [0.05 * np.exp(0.3j), 0.02])
    assert status == "Failed" and len(h) == 0


def test_reference_grid_rules():
    """m = 2*n*15 (+2k) ap, 10*n (+2k) qp, 15*n | 30*n (+2k) linprog, 30*n (+2k) qprog_phs."""
    assert assemble.assemble_fir_ap_cvx(10, [0, 0.2, 0.4, 1], [1, 1, 0, 0], [0.1, 0.1], 0.1, 1)["meta"]["m"] == 304
    P = assemble.assemble_fir_qp_cvx(10, [0, 0.2, 0.4, 1], [1, 1, 0, 0], [0.1, 0.1], 1.0, 1.0)
    assert len(P["meta"]["w"]) == 104
    assert len(assemble.assemble_fir_linprog(10, [0, 0.2, 0.4, 1], [1, 1, 0, 0], [0.1, 0.1])["meta"]["w"]) == 154
    assert len(assemble.assemble_fir_linprog(10, [-1, -0.5, 0.3, 0.8], [0, 0, 1, 1], [0.1, 0.1])["meta"]["w"]) == 304
    assert len(assemble.assemble_fir_qprog_phs(11, [0, 0.2, 0.4, 1], [1, 1, 0, 0], [0.1 * np.exp(0.2j), 0.1])["meta"]["w"]) == 334
    # lower bound floor epsilon^2 = 1e-20 and squared upper bound (fir_ap_cvx.m:105-116)
    P = assemble.assemble_fir_ap_cvx(10, [0, 0.2, 0.4, 1], [1, 1, 0, 0], [0.1, 0.1], 0.1, 1)
    assert P["meta"]["L_b"].min() == 1e-10 ** 2 and abs(P["meta"]["U_b"].max() - 1.21) < 1e-12


def test_linprog_degenerate_optimal_face_objective_and_feasibility():
    """ss/fir_linprog.m:244-252 hands the LP to linprog's active-set method, which returns a VERTEX (which one depends on
    its path from the warm start x0); an interior-point method returns the analytic centre of the optimal face.  On the
    golden cases the optimiser is unique and the two coincide.  Here the optimal face is made degenerate on purpose --
    adjacent bands leave no transition samples, so fmin = sum_tran A(i,:) = 0 (ss/fir_linprog.m:240) and the whole
    feasible polytope is optimal -- and what CAN be asserted is asserted: the oracle's point is feasible for every row
    (it is the analytic centre: strictly inside), its objective equals HiGHS's (an independent solver, itself returning
    a vertex), and the two optimisers differ (the face is not a point)."""
    from scipy.optimize import linprog
    n, f, a, d = 21, [0, 0.5, 0.5, 1.0], [1, 1, 0, 0], [0.6, 0.6]
    P = assemble.assemble_fir_linprog(n, f, a, d)
    r = conic_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"])
    assert r["status"] == conic_ipm.STATUS_OPTIMAL
    x = r["x"]
    assert np.abs(P["c"]).max() == 0.0
    assert (P["h"] - P["G"] @ x).min() >= 1e-3                 # analytic centre of the face: strictly interior
    hi = linprog(P["c"], A_ub=P["G"], b_ub=P["h"], bounds=[(None, None)] * len(P["c"]), method="highs",
                 options=dict(primal_feasibility_tolerance=1e-10, dual_feasibility_tolerance=1e-10))
    assert hi.status == 0
    assert abs(P["c"] @ x - hi.fun) <= 1e-9 * max(1.0, abs(hi.fun))
    assert np.abs(x - hi.x).max() >= 1e-3                     # same optimal value, different points of the face


@pytest.mark.parametrize("name", ["ap_twoband33", "ap_c13_58", "lin_real64", "qphs21"])
def test_centrality_corrector_and_end_game_leave_the_optimum_where_it_is(name):
    """Round 6: one centrality corrector per iteration (conic_ipm.CORR_*) cuts the iterations; the end game (POLISH) carries on past
    the stopping rule until the answer no longer depends on the path.  With and without the corrector: same verdict, same
    objective to 1e-10, same taps to 1e-7 (before the end game the two paths ended 1e-8 apart in x and up to 1e-6 in the taps)."""
    fn, args = CASES[name]
    r = {c: getattr(designers, fn)(*args, info=True, corrector=c) for c in (True, False)}
    assert r[True][1] == r[False][1] == "Solved"
    i1, i0 = r[True][2], r[False][2]
    assert i1["correctors"] == i1["iters"] and i0["correctors"] == 0 and 0 < i1["correctors_taken"] <= i1["correctors"]
    assert i1["iters"] < i0["iters"]
    assert abs(i1["pcost"] - i0["pcost"]) <= 1e-10 * max(1.0, abs(i0["pcost"]))
    # (fir_qprog_phs: a quadratic objective -- the solution moves with the square root of the gap: 1e-7 measured, held at 1e-6)
    assert relinf(r[True][0], r[False][0]) <= (1e-6 if fn == "fir_qprog_phs" else 1e-7)
    assert i1["relgap"] <= 1e-8 or i1["gap"] <= 1e-10                  # (the answer is an iterate that met the stopping rule)
