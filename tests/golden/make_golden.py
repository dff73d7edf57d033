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
Run:  python tests/golden/make_golden.py     (about 10 s; rewrites golden.json)
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import linprog

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
        out[name] = rec
        print(name, rec["status"], rec.get("pcost"), rec.get("highs_obj"), rec.get("highs_x_maxdiff"))
    with open(os.path.join(HERE, "golden.json"), "w") as fh:
        json.dump(out, fh, indent=0)


if __name__ == "__main__":
    main()
