"""CPU tests of the oracle's extended-precision KKT solve (oracle/ddlin.c, conic_ipm.factor_dd / kkt_solve_dd): the
double-double kernels against mpmath, the eigen form of the NT scalings against the closed form, and the solve that the
double-precision normal equations lose (H-1 dual band through fir_qp_cvx, k=120, obj=1e6: dzrf_mb.m:210-213)."""
import warnings

import numpy as np
import pytest

import mbfir
from oracle import assemble, conic_ipm, ddlin, designers

warnings.filterwarnings("ignore", category=RuntimeWarning)


def test_double_double_kernels_against_mpmath():
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 60
    rng = np.random.default_rng(7)
    n, k = 14, 5
    B = rng.standard_normal((n + 5, n))
    Hw = B.T @ B
    Hw = 0.5 * (Hw + Hw.T)
    U = rng.standard_normal((k, n))
    U[1] = U[0] * (1 + 1e-9) + 1e-10 * rng.standard_normal(n)              # nearly dependent strong rows
    X = 10.0 ** rng.uniform(10, 16, k)
    b = rng.standard_normal((n, 2))
    Hh, Hl = Hw.copy(), np.zeros_like(Hw)
    ddlin.rank_k(np.ascontiguousarray(U), X, Hh, Hl)
    d0 = np.diag(Hh).copy()
    assert ddlin.chol(Hh, Hl, 1e-28, d0) == 0
    Bh, Bl = b.copy(), np.zeros_like(b)
    ddlin.cho_solve(Hh, Hl, Bh, Bl)
    # the same in 60-digit arithmetic
    Hm = mp.matrix(Hw.tolist())
    for r in range(k):
        u = mp.matrix(U[r].tolist())
        Hm += mp.mpf(float(X[r])) * (u * u.T)
    xcols = [mp.lu_solve(Hm, mp.matrix(b[:, q].tolist())) for q in range(2)]
    xmax = max(abs(v) for xc in xcols for v in xc)
    for i in range(n):
        for q in range(2):
            got = mp.mpf(float(Bh[i, q])) + mp.mpf(float(Bl[i, q]))
            assert abs(got - xcols[q][i]) <= mp.mpf(10) ** -12 * xmax        # cond(H) ~ 1e18 leaves ~1e-14 of dd's 1e-32
    # strong components U x in double-double
    uh, ul = ddlin.rows_times(np.ascontiguousarray(U), Bh, Bl)
    # ... of the double-double x itself: exact to ~1e-30 relative to |U||x|
    xdd = mp.matrix([[mp.mpf(float(Bh[i, 0])) + mp.mpf(float(Bl[i, 0]))] for i in range(n)])
    um = mp.matrix(U.tolist()) * xdd
    scale = max(abs(float(v)) for v in U.ravel()) * float(xmax) * n
    for r in range(k):
        got = mp.mpf(float(uh[r, 0])) + mp.mpf(float(ul[r, 0]))
        assert abs(got - um[r, 0]) <= mp.mpf(10) ** -28 * scale


def test_eigen_form_of_the_nt_scaling_equals_the_closed_form():
    """W^-2 = eta^-2 (2 u u' - J) applied as sum_k lam_k e_k e_k' (no cancellation between the three scales);
    eig_apply(+1) is its inverse; with a cap every eigenvalue is min(lam, cap)."""
    rng = np.random.default_rng(3)
    cone = conic_ipm._Cone(5, 7, 9)
    e = cone.e()
    s = e + 0.3 * rng.standard_normal(cone.R) * (e == 0) + 0.5 * e * rng.random(cone.R)
    z = e + 0.3 * rng.standard_normal(cone.R) * (e == 0) + 0.5 * e * rng.random(cone.R)
    assert cone.min_residual(s) < 0 and cone.min_residual(z) < 0
    W = conic_ipm._Scaling(cone, s, z)
    V = rng.standard_normal((cone.R, 3))
    assert np.abs(W.eig_apply(V, -1) - W.inv2(V)).max() <= 1e-12 * np.abs(W.inv2(V)).max()
    assert np.abs(W.eig_apply(W.eig_apply(V, -1), +1) - V).max() <= 1e-12
    assert np.abs(W.eig_apply(s, -1) - z).max() <= 1e-12                     # W^-2 s = z  (NT point)
    dl, q3, bg = W.eig_weights()
    cap = float(np.median(q3[0]))
    Vc = W.eig_apply(V, -1, cap)
    # capped operator = W^-2 minus the excess rank-one terms
    ref = W.inv2(V)
    for c in range(cone.nq3):
        for lam, ev in ((q3[0][c], q3[3][c]), (q3[2][c], q3[4][c])):
            if lam > cap:
                o = cone.o3 + 3 * c
                ref[o:o + 3] -= (lam - cap) * np.outer(ev, ev @ V[o:o + 3])
        if q3[1][c] > cap:
            o = cone.o3 + 3 * c
            e0 = np.array([0.0, -q3[3][c][2], q3[3][c][1]]) * np.sqrt(2.0)
            ref[o:o + 3] -= (q3[1][c] - cap) * np.outer(e0, e0 @ V[o:o + 3])
    ref[: cone.l] = np.minimum(dl, cap)[:, None] * V[: cone.l]
    for lam, ev in ((bg[0][0], bg[3][0]), (bg[2][0], bg[4][0])):
        if lam > cap:
            ref[cone.ob:] -= (lam - cap) * np.outer(ev, ev @ V[cone.ob:])
    assert np.abs(Vc - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


def test_extended_precision_solve_changes_nothing_where_double_precision_is_enough(golden):
    """Golden quadratic-phase cases: the oracle with the extended-precision solve (its default for fir_qp_cvx) equals the
    committed vectors, which were produced before it existed."""
    from conftest import CASES
    for name in ("qp_modelA48", "qp_modelB25"):
        fn, args = CASES[name]
        h, status, info = designers.fir_qp_cvx(*args, info=True)
        g = golden[name]
        hg = np.array(g["h"]["re"]) + 1j * np.array(g["h"]["im"])
        assert status == g["status"] == "Solved"
        assert np.abs(h - hg).max() <= 1e-8 * np.abs(hg).max()
        assert abs(info["pcost"] - g["pcost"]) <= 1e-9 * max(1.0, abs(g["pcost"]))


def test_h1_dual_band_qp_form_where_double_precision_hits_the_wall():
    """H-1 dual-band saturation spec (specsat_H1_dualband.m:5-32) through fir_qp_cvx with dzrf_mb's k=120, obj=1e6 at
    n=192, grid 1536: the nearly active error cones carry NT weights ~1e16 x the rest, the double-precision normal
    equations end 'numerical' at relgap 6e-4; with the extended-precision KKT solve the same iteration reaches full
    accuracy.  (The n=384 and n=512 instances are GPU tests against a committed fixture.)"""
    n, m = 192, 1536
    f, a, d = mbfir.spec.spec_h1_dualband(n)
    P = assemble.assemble_fir_qp_cvx(n, f, a, d, 120.0, 1e6, m)
    plain = conic_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"])
    # (how far the plain solve gets before its numerical exit moves with the BLAS thread count -- relgap 6e-4 or 4e-6 with dres 2e2 --:
    #  what is asserted is the exit and that not even the reduced-accuracy rule is met)
    assert plain["status"] == conic_ipm.STATUS_NUMERICAL and (plain["relgap"] > 1.22e-4 or max(plain["pres"], plain["dres"]) > 1e-6)
    # both forms of the extended-precision solve: the double-double factorisation of the whole normal matrix and (round 4) the
    # capacitance form in plain double -- strong directions as nearly-equality constraints, a k x k Schur complement
    sol = {}
    for form in ("dd", "cap"):
        r = conic_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"], ddkkt=dict(theta=designers.DDKKT_THETA, form=form))
        assert r["status"] == conic_ipm.STATUS_OPTIMAL and r["chol_fixes"] == 0, form
        assert r["pres"] <= 1e-8 and r["dres"] <= 1e-8 and r["relgap"] <= 1e-8
        # independent certificate from the returned point
        x, s, z = r["x"], r["s"], r["z"]
        cone = conic_ipm._Cone(P["l"], P["nq3"], P["big"])
        assert np.linalg.norm(P["G"] @ x + s - P["h"]) <= 1e-8 * max(1.0, np.linalg.norm(P["h"]))
        assert np.linalg.norm(P["G"].T @ z + P["c"]) <= 1e-8 * max(1.0, np.linalg.norm(P["c"]))
        assert cone.min_residual(s) <= 1e-12 and cone.min_residual(z) <= 1e-7
        assert abs(P["c"] @ x + P["h"] @ z) <= 1e-7 * abs(P["c"] @ x)
        sol[form] = r
    assert sol["dd"]["iters"] == sol["cap"]["iters"]
    assert abs(sol["dd"]["pcost"] - sol["cap"]["pcost"]) <= 1e-10 * abs(sol["dd"]["pcost"])
    xd, xc = sol["dd"]["x"], sol["cap"]["x"]
    assert np.abs(xd[:2 * n] - xc[:2 * n]).max() <= 1e-6 * np.abs(xd[:2 * n]).max()          # (measured 3e-8; the tap criterion is 1e-6)
    h, status = designers.fir_qp_cvx(n, f, a, d, 120.0, 1e6, grid_m=m)       # the designer uses the capacitance form by default
    assert status == "Solved" and np.abs(h - (xc[:n] + 1j * xc[n:2 * n])).max() <= 1e-12
