"""The fir_qp.m probe that the device solver loses against the oracle: fir_ap_cvx(20, f, a, d, 1e5)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import numpy as np
import conftest  # noqa: F401
import mbfir
from oracle import designers
f, a, d = [-0.25, 0.25, 0.45, 1.0], [0.15, 0.15, 0, 0], [0.004, 0.002]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for dense in (0, 1):
    print("---- device, dense_trig=%d" % dense, flush=True)
    h, s, i = mbfir.fir_ap_cvx(n, f, a, d, 1e5, info=True, opts=mbfir.make_opts(verbose=1, dense_trig=dense))
    print(s, {k: i[k] for k in ("rc", "iters", "relgap", "pres", "dres", "pcost")}, flush=True)
print("---- oracle", flush=True)
ho, so, io = designers.fir_ap_cvx(n, f, a, d, 1e5, info=True, verbose=True)
print(so, io["status"], io["iters"])
