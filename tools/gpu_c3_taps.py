"""The headline instance against its fixture: the achieved ||dx|| (device vs tests/golden/c3_golden.json), the tap difference,
and what the measured amplification (tests/golden/c3_sensitivity.json) says that ||dx|| supports -- at the default tolerances and
with the tolerances tightened (gap 1e-10)."""
import json, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from conftest import c13
g = json.load(open(os.path.join(ROOT, "tests/golden/c3_golden.json")))["c3_ap_512_16384"]
s = json.load(open(os.path.join(ROOT, "tests/golden/c3_sensitivity.json")))
n = 512
f, a, d = c13(n, "duration")
xg = np.array(g["x"]); hg = np.array(g["h_re"]) + 1j * np.array(g["h_im"])
for label, kw in (("default tolerances", {}), ("abstol 1e-12 reltol 1e-10", dict(abstol=1e-12, reltol=1e-10)),
                  ("abstol 1e-13 reltol 1e-11 feastol 1e-9", dict(abstol=1e-13, reltol=1e-11, feastol=1e-9))):
    for dense in (0, 1):
        h, st, info = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=mbfir.make_opts(grid_m=16384, dense_trig=dense, **kw), info=True)
        z = mbfir.get_context().last_solution(info["n_unknowns"])
        dxr = np.abs(z[:2 * n - 1] - xg[:2 * n - 1]).max() / np.abs(xg[:2 * n - 1]).max()
        dhr = np.abs(h - hg).max() / np.abs(hg).max() if st == "Solved" else float("nan")
        print("%-40s %s %s %3d it relgap %.1e  dx_rel %.2e  dh_rel %.2e  (amplification %.2e; worst-case measured %.2e -> supports %.1e)" % (
            label, "dense  " if dense else "lattice", st, info["iters"], info["relgap"], dxr, dhr, dhr / dxr, s["amplification_max"], s["amplification_max"] * dxr), flush=True)
        if "highs_x" in s:
            xh = np.array(s["highs_x"]); hh = np.array(s["highs_h_re"]) + 1j * np.array(s["highs_h_im"])
            dxh = np.abs(z[:2 * n - 1] - xh[:2 * n - 1]).max() / np.abs(xh[:2 * n - 1]).max()
            dhh = np.abs(h - hh).max() / np.abs(hh).max()
            print("%-40s         against HiGHS's optimum: dx_rel %.2e  dh_rel %.2e  (supports %.1e)" % ("", dxh, dhh, s["amplification_max"] * dxh), flush=True)
