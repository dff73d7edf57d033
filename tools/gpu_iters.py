import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from gpu_lanes import jobs_for
n, m, count = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
jobs = jobs_for(n, count)
res = mbfir.solve_batch(jobs, streams=4, info=True, opts=mbfir.make_opts(grid_m=m, lanes=8))
its = np.array([r[2]["iters"] for r in res]).reshape(-1, 16)
print("rows = ripple index j, columns = Peak index p")
print(its)
