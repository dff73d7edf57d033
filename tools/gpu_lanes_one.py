import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from gpu_lanes import jobs_for
n, m, count, lanes, streams = [int(v) for v in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 1
jobs = jobs_for(n, count)
o = mbfir.make_opts(grid_m=m, lanes=lanes)
mbfir.solve_batch(jobs[:max(streams * max(lanes, 1), 1)], streams=streams, opts=o)
for _ in range(reps):
    t = time.time()
    res = mbfir.solve_batch(jobs, streams=streams, info=True, opts=o)
    dt = time.time() - t
    print("n %d m %d: %d designs lanes %d streams %d: %.3f s = %.1f designs/s iters %d..%d ms_chol %.1f ms_gram %.1f ms_solve %.1f" % (
        n, m, count, lanes, streams, dt, count / dt, min(r[2]["iters"] for r in res), max(r[2]["iters"] for r in res),
        res[0][2]["ms_chol"], res[0][2]["ms_gram"], res[0][2]["ms_solve"]), flush=True)
if os.environ.get("MBFIR_PRINT_ITERS"):
    print("iters per design:", [r[2]["iters"] for r in res], flush=True)
