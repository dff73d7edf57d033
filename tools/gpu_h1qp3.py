import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from oracle import designers
from conftest import relinf
n = 260
f, a, d = mbfir.spec.spec_h1_dualband(n)
ho, so, io = designers.fir_qp_cvx(n, f, a, d, 120.0, 1e6, grid_m=1000, info=True) if "info" in designers.fir_qp_cvx.__code__.co_varnames else (*designers.fir_qp_cvx(n, f, a, d, 120.0, 1e6, grid_m=1000), None)
print("oracle", so, io and {k: io[k] for k in ("iters", "pcost", "relgap", "pres", "dres") if k in io})
for dense in (0, 1):
    for rt in (0.0, 1e-10, 1e-12):
        o = mbfir.make_opts(grid_m=1000, dense_trig=dense, reltol=rt, abstol=(1e-14 if rt else 0.0))
        h, s, i = mbfir.fir_qp_cvx(n, f, a, d, 120.0, 1e6, opts=o, info=True)
        print("gpu dense=%d reltol=%g: %s it %d pcost %.12e relgap %.1e pres %.1e dres %.1e  taps vs oracle %.2e" % (dense, rt, s, i["iters"], i["pcost"], i["relgap"], i["pres"], i["dres"], relinf(h, ho) if len(h) else -1))
