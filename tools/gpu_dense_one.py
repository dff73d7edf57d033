import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import mbfir
n, m = int(sys.argv[1]), int(sys.argv[2])
f, a, d = mbfir.spec.spec_c13_bssfp(n)
h, s, i = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=mbfir.make_opts(grid_m=m, dense_trig=1), info=True)
print(s, i["iters"], i["ms_total"], i["ms_gram"], i["gram_launches"])
