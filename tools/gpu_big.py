"""Larger configurations on the GPU (BASELINE configs 2, 3-qp-form, 5)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import mbfir
from conftest import c13
which = sys.argv[1:]
def run(name, fn, *args, **kw):
    t0 = time.time()
    h, s, i = fn(*args, info=True, **kw)
    print("%-28s %s it %3d pcost %.9e pres %.1e dres %.1e relgap %.1e | %.2f s (gram %.0f ms chol %.0f ms) N=%d Mf=%d R=%d" % (
        name, s, i["iters"], i["pcost"], i["pres"], i["dres"], i["relgap"], time.time() - t0, i["ms_gram"], i["ms_chol"],
        i["n_unknowns"], i["n_freq"], i["n_rows"]), flush=True)
if not which or "c2" in which:
    f, a, d = c13(200, "duration"); run("C2 ap n=200 m=4096", mbfir.fir_ap_cvx, 200, f, a, d, 0.1, 1e-3, opts=mbfir.make_opts(grid_m=4096))
    run("C2 ap n=200 ref grid", mbfir.fir_ap_cvx, 200, f, a, d, 0.1, 1e-3)
if not which or "qp" in which:
    fq = [-0.5, -0.3, -0.1, 0.1, 0.3, 0.5]; aq = [0, 0, 1, 1, 0, 0]; dq = [0.02, 0.02, 0.02]
    run("qp n=256 ref grid", mbfir.fir_qp_cvx, 256, fq, aq, dq, 40.0, 1e3)
    run("qp n=512 m=16384", mbfir.fir_qp_cvx, 512, fq, aq, dq, 120.0, 1e6, opts=mbfir.make_opts(grid_m=16384))
if not which or "lin" in which:
    run("linprog n=511 cplx", mbfir.fir_linprog, 511, [-1, -0.4, -0.2, 0.3, 0.5, 1], [0, 0, 1, 0.8, 0, 0], [0.001, 0.002, 0.001])
    run("qprog_phs n=255", mbfir.fir_qprog_phs, 255, [-0.6, -0.3, -0.1, 0.1, 0.3, 0.6], [0, 0, 1, 1, 0, 0], [0.005, 0.01 * np.exp(0.3j), 0.005])
if "big" in which:
    f, a, d = c13(2048, "duration"); run("S-BIG ap n=2048 m=131072", mbfir.fir_ap_cvx, 2048, f, a, d, 0.1, 1e-3, opts=mbfir.make_opts(grid_m=131072, verbose=1))
