"""Single-design latency of the headline instance (and config 5) with the per-step fused factorisation and the single launch."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import mbfir
from conftest import c13
for n, m in ((512, 16384), (1024, 32768), (2048, 131072)):
    f, a, d = c13(n, "duration")
    o = mbfir.make_opts(grid_m=m)
    for split in ("0", "1", "4", ""):
        if split: os.environ["MBFIR_CHOL_SPLIT"] = split
        else: os.environ.pop("MBFIR_CHOL_SPLIT", None)
        mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=o)
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            h, s, info = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=o, info=True)
            ts.append((time.perf_counter() - t) * 1e3)
        print("n %d m %d MBFIR_CHOL_SPLIT=%s: %.1f ms (best of 3), %d iterations, ms_chol %.1f, %s" % (n, m, split, min(ts), info["iters"], info["ms_chol"], s), flush=True)
