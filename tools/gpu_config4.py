"""BASELINE config 4 on one GPU: a sweep of independent n=200, m=4096 designs (bSSFP_pulse_diff_Peak.m varies Peak)
through mbfir.solve_batch; designs/s against the number of streams."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import numpy as np
if len(sys.argv) > 1: os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]
import conftest  # noqa: F401
import mbfir
n = 200
f, a, d = mbfir.spec.spec_c13_bssfp(n)
peaks = np.logspace(-3.3, -2, 64)
jobs = [("fir_ap_cvx", (n, f, a, d, 0.1, float(p))) for p in peaks]
o = mbfir.make_opts(grid_m=4096)
mbfir.solve_batch(jobs[:8], opts=o, streams=4)
for streams in ((1, 2, 4, 6, 8) if len(sys.argv) < 2 else (4, 8, 12, 16)):
    t0 = time.time()
    res = mbfir.solve_batch(jobs, opts=o, streams=streams, info=True)
    t = time.time() - t0
    ok = sum(1 for h, s, i in res if s == "Solved")
    its = np.mean([i["iters"] for h, s, i in res])
    print("streams %d: %d designs in %.2f s = %.1f designs/s  (%d solved, %.0f iterations on average)" % (streams, len(jobs), t, len(jobs) / t, ok, its), flush=True)
