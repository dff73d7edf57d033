import os, sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np, mbfir
n, m = 512, 16384
f, a, d = mbfir.spec.spec_h1_dualband(n)
q = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dq = [x * (1.0 + 0.02 * q) for x in d]
for form in ("cap", "dd"):
    os.environ["MBFIR_DDFORM"] = form
    h, s, i = mbfir.fir_qp_cvx(n, f, a, dq, 120.0, 1e6, opts=mbfir.make_opts(grid_m=m, verbose=1), info=True)
    print(form, s, i["iters"], i["dd_iters"], i["dd_kmax"], i["pcost"], i["relgap"], i["pres"], i["dres"], flush=True)
