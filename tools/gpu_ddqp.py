import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
n = int(sys.argv[1]); m = int(sys.argv[2]); verbose = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f, a, d = mbfir.spec.spec_h1_dualband(n)
for dense in (0, 1):
    t = time.time()
    h, s, i = mbfir.fir_qp_cvx(n, f, a, d, 120.0, 1e6, opts=mbfir.make_opts(grid_m=m, verbose=verbose, dense_trig=dense), info=True)
    print("n %d m %d dense %d: %s it %d pcost %.12e relgap %.1e pres %.1e dres %.1e dd_iters %d kmax %d lattice %d  %.2f s (solve %.1f ms chol %.1f ms)" % (
        n, m, dense, s, i["iters"], i["pcost"], i["relgap"], i["pres"], i["dres"], i["dd_iters"], i["dd_kmax"], i["lattice"], time.time() - t, i["ms_solve"], i["ms_chol"]), flush=True)
