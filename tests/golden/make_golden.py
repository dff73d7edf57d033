"""Generate tests/golden/golden.json -- the committed golden vectors.

The reference ships no fixtures and cannot be executed here (MATLAB + CVX + Optimization
Toolbox), so the vectors are produced by
  (1) the oracle (oracle/, a NumPy restatement of the reference's assembly + a dense conic
      IPM): status, optimal objective, conic solution x and taps h of every case in
      tests/conftest.py:CASES;
  (2) an INDEPENDENT solver, scipy.optimize.linprog(method="highs") with 1e-10 feasibility
      tolerances, on every instance that is a pure LP (fir_linprog; fir_ap_cvx whenever the
      HiGHS optimum leaves every spike cone strictly inactive, in which case the LP optimum
      is the SOCP optimum): objective `highs_obj`, and x where the optimiser is unique.
  (3) for the programs with ACTIVE cones, which no LP solver covers, three further methods that share nothing with a
      primal-dual interior-point iteration (`pin` in the record: method, objective, max |x - x_oracle|, worst cone
      violation of the pin's own point):
        * fir_ap_cvx with active spike cones: Kelley's cutting planes on HiGHS -- every Q3 cone replaced by tangent
          half-planes, added where the LP optimum violates the cone, until the violation is <= 1e-10; every LP optimum
          is a lower bound of the conic optimum, the oracle's feasible point an upper bound;
        * fir_qprog_phs (min 1/2 x'x s.t. A x <= B, ss/fir_qprog_phs.m:339-342): Lawson & Hanson's least-distance
          programming on scipy.optimize.nnls (their active-set NNLS);
        * fir_qp_cvx (Q3 cones and the energy cone (E; x), fir_qp_cvx.m:150-165): Kraft's SLSQP (sequential least-squares
          quadratic programming, scipy.optimize.minimize) on the smooth form s0^2 - |s1|^2 >= 0, s0 >= 0, started from a
          few cutting-plane rounds on HiGHS (not from the oracle's point).
Run:  python tests/golden/make_golden.py     (about 2 min; rewrites golden.json)
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import linprog, minimize, nnls

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import CASES  # noqa: E402
from oracle import assemble, designers  # noqa: E402


def cplx(v):
    v = np.asarray(v)
    return {"re": [float(t) for t in v.real], "im": [float(t) for t in np.imag(v)]}


def highs(P):
    l = P["l"]
    r = linprog(P["c"], A_ub=P["G"][:l], b_ub=P["h"][:l], bounds=[(None, None)] * len(P["c"]), method="highs",
                options=dict(primal_feasibility_tolerance=1e-10, dual_feasibility_tolerance=1e-10))
    return r


def _cones(P):
    l, nq3, big = P["l"], P["nq3"], P.get("big", 0)
    cones = [(l + 3 * q, 3) for q in range(nq3)]
    if big:
        cones.append((l + 3 * nq3, big))
    return cones


def cone_violation(P, z):
    s = P["h"] - P["G"] @ z
    v = max([np.linalg.norm(s[r0 + 1:r0 + dim]) - s[r0] for r0, dim in _cones(P)], default=0.0)
    return float(max(v, -s[:P["l"]].min() if P["l"] else 0.0))


def cutting_plane(P, tol=1e-10, max_rounds=60):
    """Kelley's cutting planes on HiGHS.  Returns the last LP optimum (a lower bound of the conic optimum), its point
    and its worst cone violation; converges in a handful of rounds when all cones are two-dimensional (Q3)."""
    c, G, h, l = P["c"], P["G"], P["h"], P["l"]
    N = len(c)
    cones = _cones(P)

    def cut(r0, dim, U):      # u'(hb - Gb z) <= h0 - g0 z  for every unit row u of U
        return G[r0][None, :] - U @ G[r0 + 1:r0 + dim], h[r0] - U @ h[r0 + 1:r0 + dim]

    A, b = [G[:l]], [h[:l]]
    for r0, dim in cones:          # initial outer approximation: an octagon per Q3 cone, s0 >= |s_i| for the big one
        U = (np.array([[np.cos(t), np.sin(t)] for t in np.arange(8) * np.pi / 4]) if dim == 3
             else np.vstack([np.eye(dim - 1), -np.eye(dim - 1)]))
        a_, b_ = cut(r0, dim, U)
        A.append(a_)
        b.append(b_)
    A, b = np.vstack(A), np.concatenate(b)
    best = None
    for rnd in range(max_rounds):
        r = linprog(c, A_ub=A, b_ub=b, bounds=[(None, None)] * N, method="highs",
                    options=dict(primal_feasibility_tolerance=1e-10, dual_feasibility_tolerance=1e-10))
        if r.status != 0:
            break
        s = h - G @ r.x
        newA, newb, viol = [], [], 0.0
        for r0, dim in cones:
            sb = s[r0 + 1:r0 + dim]
            nb = np.linalg.norm(sb)
            viol = max(viol, nb - s[r0])
            if nb - s[r0] > 0.1 * tol and nb > 0:
                a_, b_ = cut(r0, dim, (sb / nb)[None, :])
                newA.append(a_)
                newb.append(b_)
        best = dict(obj=float(r.fun), x=r.x, viol=float(viol), rounds=rnd + 1)
        if viol <= tol:
            break
        A, b = np.vstack([A] + newA), np.concatenate([b] + newb)
    return best


def ldp_nnls(A, B):
    """min ||x|| s.t. A x <= B: Lawson & Hanson, Solving Least Squares Problems, ch. 23 (LDP) on their NNLS."""
    E = np.vstack([-A.T, -B[None, :]])
    f = np.zeros(E.shape[0])
    f[-1] = 1.0
    u, _ = nnls(E, f, maxiter=20 * E.shape[1])
    r = E @ u - f
    return -r[:-1] / r[-1]


def slsqp(P, z0):
    """SLSQP on  min c'z  s.t.  LP rows >= 0,  s0 >= 0,  s0^2 - |s1|^2 >= 0  per cone (analytic Jacobians)."""
    c, G, h, l = P["c"], P["G"], P["h"], P["l"]
    cones = _cones(P)
    heads = np.array([r0 for r0, _ in cones])

    def cons(z):
        s = h - G @ z
        q = np.array([s[r0] ** 2 - s[r0 + 1:r0 + dim] @ s[r0 + 1:r0 + dim] for r0, dim in cones])
        return np.concatenate([s[:l], s[heads], q])

    def jac(z):
        s = h - G @ z
        Q = np.array([-2 * s[r0] * G[r0] + 2 * s[r0 + 1:r0 + dim] @ G[r0 + 1:r0 + dim] for r0, dim in cones])
        return np.vstack([-G[:l], -G[heads], Q])

    r = minimize(lambda z: c @ z, z0, jac=lambda z: c, constraints=[dict(type="ineq", fun=cons, jac=jac)], method="SLSQP",
                 options=dict(maxiter=2000, ftol=1e-15))
    return r.x


def pin_record(P, z, x_oracle, method, **extra):
    return dict(method=method, obj=float(P["c"] @ z), x_maxdiff=float(np.abs(z - x_oracle).max()), viol=cone_violation(P, z), **extra)


def main():
    out = {}
    for name, (fn, args) in CASES.items():
        h, status, info = getattr(designers, fn)(*args, info=True)
        rec = {"designer": fn, "status": status, "iters": int(info["iters"])}
        if status == "Solved":
            rec.update(pcost=float(info["pcost"]), gap=float(info["gap"]), pres=float(info["pres"]),
                       dres=float(info["dres"]), x=[float(t) for t in info["x"]], h=cplx(h))
        if fn in ("fir_linprog", "fir_ap_cvx") and status == "Solved":
            P = (assemble.assemble_fir_linprog(*args) if fn == "fir_linprog" else assemble.assemble_fir_ap_cvx(*args))
            r = highs(P)
            if r.status == 0:
                ok = True
                if fn == "fir_ap_cvx":      # the LP relaxation only pins the SOCP when every cone is slack
                    q = (P["h"][P["l"]:] - P["G"][P["l"]:] @ r.x).reshape(-1, 3)
                    ok = bool((q[:, 0] - np.hypot(q[:, 1], q[:, 2])).min() > 1e-9)
                if ok:
                    rec["highs_obj"] = float(r.fun)
                    rec["highs_x_maxdiff"] = float(np.abs(r.x - info["x"]).max())
        if status == "Solved" and "highs_obj" not in rec and fn != "fir_linprog":
            P = getattr(assemble, "assemble_" + fn)(*args)
            xo = np.asarray(info["x"])
            if fn == "fir_ap_cvx":
                cp = cutting_plane(P)
                rec["pin"] = pin_record(P, cp["x"], xo, "kelley cutting planes on HiGHS", rounds=cp["rounds"], lower_bound=cp["obj"])
            elif fn == "fir_qprog_phs":
                x = ldp_nnls(P["meta"]["A"], P["meta"]["B"])
                rec["pin"] = pin_record(P, np.append(x, np.linalg.norm(x)), xo, "least-distance programming on NNLS (Lawson-Hanson)")
            else:
                cp = cutting_plane(P, max_rounds=12)      # a starting point near the optimum, from HiGHS alone
                z0 = cp["x"] if cp else np.append(np.zeros(len(xo) - 3), [10.0, 10.0, 10.0])
                z = slsqp(P, z0)
                rec["pin"] = pin_record(P, z, xo, "SLSQP from a cutting-plane start", lower_bound=cp["obj"] if cp else None)
        out[name] = rec
        print(name, rec["status"], rec.get("pcost"), rec.get("highs_obj"), rec.get("highs_x_maxdiff"), rec.get("pin"))
    with open(os.path.join(HERE, "golden.json"), "w") as fh:
        json.dump(out, fh, indent=0)


if __name__ == "__main__":
    main()
