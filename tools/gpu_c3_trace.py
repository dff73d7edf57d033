"""Verbose IPM trace of the headline design (S-C13, n=512, m=16384, fir_ap_cvx form)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import conftest  # noqa: F401
import mbfir
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
f, a, d = mbfir.spec.spec_c13_bssfp(n)
h, s, i = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, info=True, opts=mbfir.make_opts(grid_m=m, verbose=1))
print(s, i["iters"], i["ms_solve"])
