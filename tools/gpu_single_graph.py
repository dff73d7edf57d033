"""Single-design latency with and without the per-iteration launch graphs (MBFIR_GRAPH), and equality of the results."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from conftest import c13, CASES
cases = [("fir_ap_cvx n=64", "fir_ap_cvx", (64,) + tuple(c13(64)) + (0.1, 1e-3), {}),
         ("fir_ap_cvx n=200 m=4096", "fir_ap_cvx", (200,) + tuple(c13(200, "duration")) + (0.1, 1e-3), dict(grid_m=4096)),
         ("fir_ap_cvx n=260 H-1", "fir_ap_cvx", (260,) + tuple(mbfir.spec.spec_h1_dualband(260)) + (0.1, 1e-3), {}),
         ("fir_ap_cvx n=512 m=16384", "fir_ap_cvx", (512,) + tuple(c13(512, "duration")) + (0.1, 1e-3), dict(grid_m=16384)),
         ("fir_linprog lin_real64", "fir_linprog", CASES["lin_real64"][1], {}),
         ("fir_qprog_phs qphs21", "fir_qprog_phs", CASES["qphs21"][1], {}),
         ("fir_ap_cvx n=2048 m=131072", "fir_ap_cvx", (2048,) + tuple(c13(2048, "duration")) + (0.1, 1e-3), dict(grid_m=131072))]
for label, fn, args, okw in cases:
    out = {}
    for mode in ("0", "1"):
        os.environ["MBFIR_GRAPH"] = mode
        o = mbfir.make_opts(**okw)
        getattr(mbfir, fn)(*args, opts=o)
        best = 1e9
        for _ in range(3):
            t = time.time(); h, s, i = getattr(mbfir, fn)(*args, opts=o, info=True); best = min(best, time.time() - t)
        out[mode] = (h, s, i, best)
    (h0, s0, i0, t0), (h1, s1, i1, t1) = out["0"], out["1"]
    same = s0 == s1 and i0["iters"] == i1["iters"] and i0["pcost"] == i1["pcost"] and np.array_equal(h0, h1)
    print("%-28s eager %.1f ms, graphs %.1f ms (%.2fx)  %s %d iterations; results %s; chol %.1f / %.1f ms, normal matrix %.1f / %.1f ms, builds %d / %d" % (
        label, t0 * 1e3, t1 * 1e3, t0 / t1, s1, i1["iters"], "identical" if same else "DIFFER", i0["ms_chol"], i1["ms_chol"], i0["ms_gram"], i1["ms_gram"], i0["builds"], i1["builds"]), flush=True)
os.environ.pop("MBFIR_GRAPH", None)
