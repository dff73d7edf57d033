"""Verbose IPM trace of the headline design (stderr of the solver)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mbfir
from conftest import c13
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
f, a, d = c13(n, "duration")
h, s, i = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=mbfir.make_opts(grid_m=m, verbose=1), info=True)
print(s, i["iters"], i["pcost"])
