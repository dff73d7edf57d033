"""Per-iteration trace (opts.verbose, stderr) of BASELINE config 3's family: fir_qp_cvx, H-1 dual band, k=120, obj=1e6."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import mbfir
n, m = int(sys.argv[1]), int(sys.argv[2])
f, a, d = mbfir.spec.spec_h1_dualband(n)
h, s, i = mbfir.fir_qp_cvx(n, f, a, d, 120.0, 1e6, opts=mbfir.make_opts(grid_m=m, verbose=1), info=True)
print(s, i["iters"], i["dd_iters"], i["correctors_taken"], i["correctors"], "pcost %.12e relgap %.2e" % (i["pcost"], i["relgap"]))
