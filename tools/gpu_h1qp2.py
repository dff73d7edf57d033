import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mbfir
for n, m, k, obj in ((260, 0, 120.0, 1e6), (384, 4096, 120.0, 1e6), (512, 2048, 120.0, 1e6), (512, 5120, 120.0, 1e6), (512, 16384, 120.0, 1e6), (512, 16384, 120.0, 1e3), (512, 16384, 120.0, 1.0)):
    f, a, d = mbfir.spec.spec_h1_dualband(n)
    h, s, i = mbfir.fir_qp_cvx(n, f, a, d, k, obj, opts=mbfir.make_opts(grid_m=m), info=True)
    print("n=%d m=%d k=%g obj=%g: %s it %d pcost %.6e pres %.1e dres %.1e relgap %.1e" % (n, m, k, obj, s, i["iters"], i["pcost"], i["pres"], i["dres"], i["relgap"]), flush=True)
