"""Determinism probe: the same solve repeated in fresh and reused contexts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import mbfir
from conftest import c13
f, a, d = c13(64)
def run(ctx, n):
    h, s, i = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, ctx=ctx, info=True)
    print(n, s, "rc", i["rc"], "it", i["iters"], "pcost %.12e dres %.2e pres %.2e" % (i["pcost"], i["dres"], i["pres"]), "hsum", np.abs(h).sum() if len(h) else None, flush=True)
for rep in range(3):
    ctx = mbfir.Context(0); run(ctx, 58); ctx.close()
ctx = mbfir.Context(0)
for n in (58, 50, 58, 64, 58):
    run(ctx, n)
