"""fir_qp.m transition search on the device next to the oracle, probe by probe (verdict + solver status)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import numpy as np
import conftest  # noqa: F401
import mbfir
from oracle import designers
f, a, d = [-0.25, 0.25, 0.45, 1.0], [0.15, 0.15, 0, 0], [0.004, 0.002]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
def both(nn, ff):
    hg, sg, ig = mbfir.fir_ap_cvx(nn, ff, a, d, 1e5, info=True)
    ho, so, io = designers.fir_ap_cvx(nn, ff, a, d, 1e5, info=True)
    print("n=%d df=%.6f | gpu %-6s rc %d it %3d relgap %.1e pres %.1e dres %.1e pcost %.6e | oracle %-6s st %d it %3d relgap %.1e pres %.1e dres %.1e pcost %.6e" % (
        nn, (ff[2] - ff[1]) / 2, sg, ig["rc"], ig["iters"], ig["relgap"], ig["pres"], ig["dres"], ig["pcost"], so, io["status"], io["iters"], io["relgap"], io["pres"], io["dres"], io["pcost"]), flush=True)
    return ho, so
mbfir.fir_qp(n, f, a, d, 0, 0.5, designer=both)
