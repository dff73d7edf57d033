"""What splitting the slow unit of the headline batch over idle streams could buy: the sixteen designs of the
bench's tightest ripple pair (82-92 iterations) as one lock-step unit of 16 on one stream, two units of 8 on two, four units of 4 on
four, eight of 2 on eight -- wall-clock for the sixteen designs, i.e. the length of the batch's tail if it were run that way."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
import bench
n, m = 512, 16384
jobs = bench.sweep_jobs(mbfir, n, 16)            # j = 0: the tight pair, 16 Peak values
ctxs = [mbfir.Context(0) for _ in range(8)]
for lanes, streams in ((16, 1), (8, 2), (4, 4), (2, 8)):
    o = mbfir.make_opts(grid_m=m, lanes=lanes)
    mbfir.solve_batch(jobs, ctxs=ctxs[:streams], opts=o)
    best = 1e9
    for _ in range(3):
        t = time.time(); res = mbfir.solve_batch(jobs, ctxs=ctxs[:streams], opts=o, info=True); best = min(best, time.time() - t)
    its = [r[2]["iters"] for r in res]
    print("16 slow designs as %2d unit(s) of %2d on %d stream(s): %.1f ms (%d-%d iterations; %.2f ms per iteration of the longest)" % (
        16 // lanes, lanes, streams, best * 1e3, min(its), max(its), best * 1e3 / max(its)), flush=True)
