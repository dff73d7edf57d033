"""Lattice (matrix-free) mode vs the dense path on the GPU: same verdicts, taps, iteration counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mbfir
from conftest import c13, CASES
def rel(a, b):
    if len(a) == 0 and len(b) == 0: return 0.0
    if len(a) != len(b): return float("inf")
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
for name, (which, args) in CASES.items():
    fn = getattr(mbfir, which)
    out = []
    for dense in (1, 0):
        t0 = time.time()
        h, s, i = fn(*args, info=True, opts=mbfir.make_opts(dense_trig=dense))
        out.append((h, s, i, time.time() - t0))
    (hd, sd, idn, td), (hl, sl, il, tl) = out
    print("%-22s dense %s it %3d %.3fs | lattice(%d) %s it %3d %.3fs | rel diff %.2e" % (name, sd, idn["iters"], td, il["lattice"], sl, il["iters"], tl, rel(hl, hd)), flush=True)
f, a, d = c13(512, "duration")
for dense in (1, 0):
    o = mbfir.make_opts(grid_m=16384, dense_trig=dense)
    mbfir.fir_ap_cvx(512, f, a, d, 0.1, 1e-3, opts=o)
    t0 = time.time(); h, s, i = mbfir.fir_ap_cvx(512, f, a, d, 0.1, 1e-3, opts=o, info=True); t = time.time() - t0
    print("C3 dense=%d: %s it %d pcost %.10e %.1f ms (gram %.1f chol %.1f)" % (dense, s, i["iters"], i["pcost"], t * 1e3, i["ms_gram"], i["ms_chol"]), flush=True)
    if dense: href = h
print("C3 taps rel diff %.2e" % rel(h, href))
