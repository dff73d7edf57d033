import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from oracle import designers
from test_bloch_gpu import c13_args
cf, rng, FA, rp = c13_args()
rf_pulse, b, rf_spec, b_spec = mbfir.dzrf_mb(100, 0.04, cf, rng, FA, rp, "ex", "ap_minorder_cvx", "C-13", 0, 1, None, 0, 58)
f, a, d = b_spec["f"], b_spec["a"], b_spec["d"]
ho, so, io = designers.fir_ap_cvx(58, f, a, d, 0.1, 1e-3, info=True)
for dense in (0, 1):
    hg, sg, ig = mbfir.fir_ap_cvx(58, f, a, d, 0.1, 1e-3, info=True, opts=mbfir.make_opts(dense_trig=dense))
    z = mbfir.get_context().last_solution(ig["n_unknowns"])
    print("chunk", os.environ.get("MBFIR_CHUNK"), "dense", dense, sg, "iters dev %d orc %d" % (ig["iters"], io["iters"]), "relgap dev %.2e orc %.2e" % (ig["relgap"], io["relgap"]),
          "pcost diff %.2e" % (abs(ig["pcost"] - io["pcost"]) / abs(io["pcost"])), "x diff %.2e" % (np.abs(z - io["x"]).max() / np.abs(io["x"]).max()),
          "taps diff %.2e" % (np.abs(hg - ho).max() / np.abs(ho).max()))
