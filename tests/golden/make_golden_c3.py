"""Generate tests/golden/c3_golden.json: the HEADLINE instance of BASELINE.json -- n=512 taps, 16384 grid points,
arbitrary-phase SOCP (fir_ap_cvx form) on the bSSFP C-13 spec in the fixed-duration regime (S-C13,
obj=0.1, Peak=1e-3) -- solved by the oracle (oracle/conic_ipm.py: a few minutes of NumPy time on 8 cores, too long for
a test, hence a fixture).  Stored: status, iteration count, objective, gap, residuals, the conic solution x*, the taps,
and two independent parts:
  * a primal-dual certificate re-evaluated from the returned (x, s, z) in plain NumPy;
  * the HiGHS optimum of the LP relaxation (scipy.optimize.linprog, spike cones dropped, 1e-10 tolerances), which IS
    the SOCP optimum when its point leaves every spike cone slack (`highs_cones_slack`): an unrelated solver's value of
    the same program.  (The dense LP is 32788 x 1024; HiGHS needs its time.)
Run:  python tests/golden/make_golden_c3.py [--no-highs | --twin-only]      (rewrites c3_golden.json; --twin-only: the second record alone)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import c13  # noqa: E402
from oracle import assemble, conic_ipm, specfact  # noqa: E402


def twin_only(n, m, P):
    G, h, c, l = P["G"], P["h"], P["c"], P["l"]
    path = os.path.join(HERE, "c3_golden.json")
    with open(path) as fh:
        rec = json.load(fh)["c3_ap_512_16384"]
    t = time.time()
    r0 = conic_ipm.solve(c, G, h, l, P["nq3"], P["big"], corrector=False)
    taps0 = specfact.fmp2(specfact.x_to_r(r0["x"][: 2 * n - 1], n))
    rec0 = dict(n=n, grid_m=m, status=int(r0["status"]), iters=int(r0["iters"]), pcost=float(r0["pcost"]), gap=float(r0["gap"]), relgap=float(r0["relgap"]),
                pres=float(r0["pres"]), dres=float(r0["dres"]), seconds=time.time() - t, x=[float(v) for v in r0["x"]],
                h_re=[float(v) for v in taps0.real], h_im=[float(v) for v in taps0.imag])
    print({k: v for k, v in rec0.items() if k not in ("x", "h_re", "h_im")}, flush=True)
    with open(path, "w") as fh:
        json.dump({"c3_ap_512_16384": rec, "c3_ap_512_16384_no_corrector": rec0}, fh)


def main():
    n, m = 512, 16384
    f, a, d = c13(n, "duration")
    P = assemble.assemble_fir_ap_cvx(n, f, a, d, 0.1, 1e-3, m)
    G, h, c, l = P["G"], P["h"], P["c"], P["l"]
    if "--twin-only" in sys.argv:                            # (keep the first record as it is on file, redo the second)
        return twin_only(n, m, P)
    t = time.time()
    r = conic_ipm.solve(c, G, h, l, P["nq3"], P["big"])
    secs = time.time() - t
    x, s, z = r["x"], r["s"], r["z"]
    cone = conic_ipm._Cone(l, P["nq3"], P["big"])
    cert = dict(pres=float(np.linalg.norm(G @ x + s - h) / max(1.0, np.linalg.norm(h))),
                dres=float(np.linalg.norm(G.T @ z + c) / max(1.0, np.linalg.norm(c))),
                s_outside=float(cone.min_residual(s)), z_outside=float(cone.min_residual(z)),
                sz=float(s @ z), pcost=float(c @ x), dcost=float(-h @ z))
    taps = specfact.fmp2(specfact.x_to_r(x[: 2 * n - 1], n))
    q = (h[l:] - G[l:] @ x).reshape(-1, 3)
    rec = dict(n=n, grid_m=m, obj=0.1, peak=1e-3, f=[float(v) for v in f], a=[float(v) for v in a], d=[float(v) for v in d],
               status=int(r["status"]), iters=int(r["iters"]), pcost=float(r["pcost"]), dcost=float(r["dcost"]), gap=float(r["gap"]),
               relgap=float(r["relgap"]), pres=float(r["pres"]), dres=float(r["dres"]), chol_fixes=int(r["chol_fixes"]), seconds=secs,
               x=[float(v) for v in x], h_re=[float(v) for v in taps.real], h_im=[float(v) for v in taps.imag], certificate=cert,
               min_spike_cone_slack=float((q[:, 0] - np.hypot(q[:, 1], q[:, 2])).min()))
    print({k: v for k, v in rec.items() if k not in ("x", "h_re", "h_im", "f", "a", "d")}, flush=True)
    if "--reuse-highs" in sys.argv:
        with open(os.path.join(HERE, "c3_golden.json")) as fh:
            oldrec = json.load(fh)["c3_ap_512_16384"]
        for k in ("highs_status", "highs_seconds", "highs_obj", "highs_cones_slack", "highs_x"):
            rec[k] = oldrec[k]
        rec["highs_x_maxdiff"] = float(np.abs(np.array(oldrec["highs_x"]) - x).max())
    elif "--no-highs" not in sys.argv:
        from scipy.optimize import linprog
        t = time.time()
        rh = linprog(c, A_ub=G[:l], b_ub=h[:l], bounds=[(None, None)] * len(c), method="highs",
                     options=dict(primal_feasibility_tolerance=1e-10, dual_feasibility_tolerance=1e-10))
        rec["highs_status"] = int(rh.status)
        rec["highs_seconds"] = time.time() - t
        if rh.status == 0:
            qh = (h[l:] - G[l:] @ rh.x).reshape(-1, 3)
            rec["highs_obj"] = float(rh.fun)
            rec["highs_cones_slack"] = bool((qh[:, 0] - np.hypot(qh[:, 1], qh[:, 2])).min() > 0)
            rec["highs_x_maxdiff"] = float(np.abs(rh.x - x).max())
            rec["highs_x"] = [float(v) for v in rh.x]        # (kept: --reuse-highs redoes the oracle's records without the five minutes of HiGHS)
        print({k: v for k, v in rec.items() if k.startswith("highs")}, flush=True)
    # Second record (round 6): the same instance WITHOUT the centrality corrector (conic_ipm.solve(corrector=False); the device:
    # MBFIR_CORRECTOR=0).  The corrector's take-or-leave decisions amplify rounding differences until device and oracle walk
    # different paths to the same optimum (they end 6e-8 apart in x here: the distance of either endpoint from the optimum); without
    # it the device follows the oracle step for step, which is what pins the KERNELS' arithmetic at this size (x to 1e-9, taps to 1e-6).
    path = os.path.join(HERE, "c3_golden.json")
    t = time.time()
    r0 = conic_ipm.solve(c, G, h, l, P["nq3"], P["big"], corrector=False)
    taps0 = specfact.fmp2(specfact.x_to_r(r0["x"][: 2 * n - 1], n))
    rec0 = dict(n=n, grid_m=m, status=int(r0["status"]), iters=int(r0["iters"]), pcost=float(r0["pcost"]), gap=float(r0["gap"]), relgap=float(r0["relgap"]),
                pres=float(r0["pres"]), dres=float(r0["dres"]), seconds=time.time() - t, x=[float(v) for v in r0["x"]],
                h_re=[float(v) for v in taps0.real], h_im=[float(v) for v in taps0.imag])
    print({k: v for k, v in rec0.items() if k not in ("x", "h_re", "h_im")}, flush=True)
    with open(path, "w") as fh:
        json.dump({"c3_ap_512_16384": rec, "c3_ap_512_16384_no_corrector": rec0}, fh)


if __name__ == "__main__":
    main()
