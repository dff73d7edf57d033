"""The headline instance (tests/golden/c3_golden.json) on the device against the fixture: where each ends (iterations, gap measures)
and how far apart the conic solutions and the taps are."""
import json, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from conftest import c13
n = 512
f, a, d = c13(n, "duration")
g = json.load(open(os.path.join(ROOT, "tests", "golden", "c3_golden.json")))["c3_ap_512_16384"]
xg = np.array(g["x"]); hg = np.array(g["h_re"]) + 1j * np.array(g["h_im"])
print("fixture: iters %d gap %.2e relgap %.2e pres %.1e dres %.1e pcost %.13e" % (g["iters"], g["gap"], g["relgap"], g["pres"], g["dres"], g["pcost"]))
for env in ({}, {"MBFIR_CORRECTOR": "0"}):
    os.environ.pop("MBFIR_CORRECTOR", None); os.environ.update(env)
    for dense in (0, 1):
        h, st, i = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=mbfir.make_opts(grid_m=16384, dense_trig=dense, verbose=int(len(sys.argv) > 1 and dense == 0)), info=True)
        z = mbfir.get_context().last_solution(i["n_unknowns"])
        print("device %s dense %d: %s iters %d (taken %d) gap %.2e relgap %.2e pres %.1e dres %.1e pcost %.13e | dx_rel %.2e taps %.2e" % (
            env, dense, st, i["iters"], i["correctors_taken"], i["gap"], i["relgap"], i["pres"], i["dres"], i["pcost"],
            np.abs(z[:2 * n - 1] - xg[:2 * n - 1]).max() / np.abs(xg).max(), np.abs(h - hg).max() / np.abs(hg).max()), flush=True)
