"""Oracle end-to-end designers (test infrastructure, see oracle/__init__.py): the four
reference entry points restated on top of assemble.py / conic_ipm.py / specfact.py, with
the reference's return convention (h, status), status in {'Solved', 'Failed'}, h empty
on failure (fir_ap_cvx.m:176-182, fir_qp_cvx.m:200-206, ss/fir_linprog.m:265-271,
ss/fir_qprog_phs.m:388-394)."""
import numpy as np

from . import assemble, conic_ipm, specfact

_OK = (conic_ipm.STATUS_OPTIMAL, conic_ipm.STATUS_OPTIMAL_INACCURATE)
_EMPTY = np.zeros(0, dtype=np.complex128)


def _solve(P, kw):
    return conic_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"], **kw)


def _ret(h, r, info):
    ok = r["status"] in _OK
    out = (h if ok else _EMPTY, "Solved" if ok else "Failed")
    if info:
        d = {k: v for k, v in r.items() if k not in ("x", "s", "z")}
        d["x"] = r["x"]
        return out + (d,)
    return out


def fir_ap_cvx(n, f, a, d, obj=0.0, Peak=1e-3, dbg=0, grid_m=0, info=False, **kw):
    P = assemble.assemble_fir_ap_cvx(n, f, a, d, obj, Peak, grid_m)
    r = _solve(P, kw)
    h = specfact.fmp2(specfact.x_to_r(r["x"][: 2 * n - 1], n)) if r["status"] in _OK else None   # :185-186,202
    return _ret(h, r, info)


DDKKT_THETA = 1e6       # csrc/api.cpp uses the same value


def fir_qp_cvx(n, f, a, d, k=100.0, obj=0.0, dbg=0, grid_m=0, info=False, **kw):
    P = assemble.assemble_fir_qp_cvx(n, f, a, d, k, obj, grid_m)
    # the nearly active error cones of this designer need the extended-precision KKT solve (conic_ipm.factor_dd): its
    # capacitance form in plain double, as on the device (capkkt.hip); ddkkt=dict(theta=..., form="dd") is the double-double one
    kw.setdefault("ddkkt", dict(theta=DDKKT_THETA, form="cap"))
    r = _solve(P, kw)
    x = r["x"]
    return _ret(x[:n] + 1j * x[n:2 * n], r, info)                                                  # :209


def fir_linprog(n, f, a, d, h0=None, dbg=0, grid_m=0, info=False, **kw):
    try:
        P = assemble.assemble_fir_linprog(n, f, a, d, grid_m)
    except assemble.EarlyFail:                                                                   # :66-75
        r = dict(status=conic_ipm.STATUS_NUMERICAL, iters=0, pcost=np.nan, x=None)
        return _ret(None, r, info)
    r = _solve(P, kw)
    m = P["meta"]
    h = assemble.fill_h_linprog(r["x"], m["nhalf"], m["real_filter"], m["odd_filter"])           # :274-296
    return _ret(h, r, info)


def fir_qprog_phs(n, f, ac, dc, x0=None, dbg=0, grid_m=0, info=False, **kw):
    try:
        P = assemble.assemble_fir_qprog_phs(n, f, ac, dc, grid_m)
    except assemble.EarlyFail:                                                                   # :193-202
        r = dict(status=conic_ipm.STATUS_NUMERICAL, iters=0, pcost=np.nan, x=None)
        return _ret(None, r, info)
    r = _solve(P, kw)
    x = r["x"]
    return _ret(x[:n] + 1j * x[n:2 * n], r, info)                                                  # :389
