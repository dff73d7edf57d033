import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mbfir
n, m = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dense = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f, a, d = mbfir.spec.spec_h1_dualband(n)
h, s, i = mbfir.fir_qp_cvx(n, f, a, d, 120.0, 1e6, opts=mbfir.make_opts(grid_m=m, dense_trig=dense, verbose=1), info=True)
print(s, i["iters"], i["pcost"], i["pres"], i["dres"])
