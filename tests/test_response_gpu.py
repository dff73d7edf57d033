"""The reference's visual checks turned into predicates: the frequency response of the returned taps
against the band spec for every designer (`check_response`, ss/fir_pm.m:212-236), and `A x` = DFT of the filled taps
(ss/fir_linprog.m:254-263, ss/fir_qprog_phs.m:376-382)."""
import numpy as np
import pytest
from conftest import CASES

import mbfir

pytestmark = pytest.mark.gpu


def response(h, w):
    k = np.arange(len(h))
    return (h[None, :] * np.exp(-1j * w[:, None] * k[None, :])).sum(1)


def check_response(h, f, a, d, slack):
    """| |H(w)| - a | <= d (1 + slack) on 400 points of every band."""
    f = np.asarray(f) * np.pi
    worst = 0.0
    for i in range(len(d)):
        w = np.linspace(f[2 * i], f[2 * i + 1], 400)
        amp = np.abs(a[2 * i]) + (np.abs(a[2 * i + 1]) - np.abs(a[2 * i])) * (w - w[0]) / max(w[-1] - w[0], 1e-300)
        dev = np.max(np.abs(np.abs(response(h, w)) - amp)) / abs(d[i])
        worst = max(worst, dev)
    assert worst <= 1 + slack, worst
    return worst


@pytest.mark.parametrize("name", [nm for nm in sorted(CASES) if "infeasible" not in nm])
def test_band_response_of_every_golden_design(name):
    which, args = CASES[name]
    h, status, info = getattr(mbfir, which)(*args, info=True)
    assert status == "Solved"
    n, f, a, d = args[:4]
    if which == "fir_ap_cvx":
        # |H|^2 = S up to fmp2's abs() folding (fir_ap_cvx.m:281) and the grid: up to 13 % of the ripple on these cases
        check_response(h, f, a, d, 0.15)
    elif which == "fir_qp_cvx":
        scale = 1.0
        if np.ndim(args[5]) > 0:                                 # model B minimises delta, the bands get D_i * delta (:176)
            scale = mbfir.get_context().last_solution(info["n_unknowns"])[2 * n]
            assert scale > 0
        check_response(h, f, a, np.asarray(d) * scale, 0.05)     # |H - Hd| <= D at the 10 n grid points implies ||H| - a| <= d there
    elif which == "fir_linprog":
        check_response(h, f, a, d, 0.02)
    else:
        check_response(h, f, np.abs(a), np.abs(np.real(d)) + np.abs(np.imag(d)), 0.05)


@pytest.mark.parametrize("name", ["lin_real64", "lin_real33", "lin_cplx31", "lin_cplx32"])
def test_linprog_amplitude_equals_A_times_x(name):
    """'Frequency response calculated with A' (ss/fir_linprog.m:254-263): the rows +A of the program applied to the
    solver's x give the zero-phase amplitude of the filled taps at the design frequencies."""
    which, args = CASES[name]
    h, status, info = mbfir.fir_linprog(*args, info=True)
    assert status == "Solved"
    n = args[0]
    z = mbfir.get_context().last_solution(info["n_unknowns"])
    rc, P = mbfir.assemble_dense(2, n, args[1], args[2], args[3])
    assert rc == 0
    Mf = P["Mf"]
    Ax = P["G"][:Mf] @ z                                         # first Mf rows: A x <= U
    amp = np.real(response(h, P["w"]) * np.exp(1j * P["w"] * (n - 1) / 2))     # linear phase removed
    assert np.max(np.abs(Ax - amp)) <= 1e-9 * max(1.0, np.max(np.abs(amp)))
    assert np.max(np.abs(np.imag(response(h, P["w"]) * np.exp(1j * P["w"] * (n - 1) / 2)))) <= 1e-9


@pytest.mark.parametrize("name", ["qphs21", "qphs22"])
def test_qprog_phs_response_equals_the_program_rows(name):
    """ss/fir_qprog_phs.m:376-382: the complex response at the design frequencies from x = [Re h; Im h]."""
    which, args = CASES[name]
    h, status, info = mbfir.fir_qprog_phs(*args, info=True)
    assert status == "Solved"
    n = args[0]
    z = mbfir.get_context().last_solution(info["n_unknowns"])
    assert np.allclose(z[:n] + 1j * z[n:2 * n], h, atol=1e-14)
    H = response(h, np.asarray(args[1]) * np.pi)
    assert np.all(np.isfinite(H))
