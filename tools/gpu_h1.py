"""BASELINE config 3: H-1 dual-band saturation spec at n=512, m=16384, arbitrary-phase and quadratic-phase forms."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mbfir
for n, m in ((260, 0), (512, 16384)):
    f, a, d = mbfir.spec.spec_h1_dualband(n)
    o = mbfir.make_opts(grid_m=m)
    for name, fn, args in (("fir_ap_cvx obj=0.1", mbfir.fir_ap_cvx, (n, f, a, d, 0.1, 1e-3)), ("fir_qp_cvx k=120 obj=1e6", mbfir.fir_qp_cvx, (n, f, a, d, 120.0, 1e6))):
        fn(*args, opts=o)
        t0 = time.time(); h, s, i = fn(*args, opts=o, info=True); t = time.time() - t0
        print("n=%d m=%d %-26s %s it %3d pcost %.8e pres %.1e dres %.1e relgap %.1e lattice %d | %.1f ms (chol %.1f)" % (n, i["n_freq"], name, s, i["iters"], i["pcost"], i["pres"], i["dres"], i["relgap"], i["lattice"], t * 1e3, i["ms_chol"]), flush=True)
