"""BASELINE config 4 (256 designs n=200 m=4096, units of 32 on 4 streams) with the centrality corrector on and off, same box."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); warnings.filterwarnings("ignore")
import numpy as np, mbfir
from gpu_lanes import jobs_for
jobs = jobs_for(200, 256)
o = mbfir.make_opts(grid_m=4096, lanes=32)
for corr in ("1", "0", "1", "0"):
    os.environ["MBFIR_CORRECTOR"] = corr
    mbfir.solve_batch(jobs, streams=4, opts=o)
    best = 1e9
    for _ in range(3):
        t = time.time(); res = mbfir.solve_batch(jobs, streams=4, info=True, opts=o); best = min(best, time.time() - t)
    print("config 4, corrector %s: %.1f ms = %.0f designs/s, solved %d, iterations %.1f" % (corr, 1e3 * best, 256 / best, sum(1 for r in res if r[1] == "Solved"), np.mean([r[2]["iters"] for r in res])), flush=True)
