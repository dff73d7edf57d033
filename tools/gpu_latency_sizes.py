"""Single-design latency of fir_ap_cvx(S-C13) over the sizes VERDICT r4 item 2 names (n = 64, 100, 200, 512 at the reference's grid rule and
the headline's grid), with the launch fusions and the speculative head on and off."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import mbfir
from conftest import c13
for n, m in ((64, 0), (100, 0), (200, 4096), (512, 16384)):
    f, a, d = c13(n, "duration") if n >= 200 else c13(n)
    o = mbfir.make_opts(grid_m=m) if m else None
    for env in ({}, {"MBFIR_FUSE": "0", "MBFIR_SPECULATE": "0"}):
        for k in ("MBFIR_FUSE", "MBFIR_SPECULATE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=o)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            h, s, info = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=o, info=True)
            ts.append((time.perf_counter() - t) * 1e3)
        print("n %4d m %6d %-40s %.2f ms (best of 5), %d iterations, %.0f us per iteration, %s" % (
            n, info["n_freq"], "round 4's sequence" if env else "round 5 (fused launches, head ahead)", min(ts), info["iters"], 1e3 * min(ts) / max(1, info["iters"]), s), flush=True)
