"""BASELINE config 3 as written at size: the 8 designs of the bench's batch as ONE lock-step unit on the extended-precision path against
the same designs solved one by one -- verdict, iterations, iterations on the path, largest strong set, objective and taps bit for bit."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
n, m = 512, 16384
f, a, d = mbfir.spec.spec_h1_dualband(n)
jobs = [("fir_qp_cvx", (n, f, a, [x * (1.0 + 0.02 * q) for x in d], 120.0, 1e6)) for q in range(8)]
ctx = mbfir.Context(0)
res = mbfir.solve_batch(jobs, ctxs=[ctx], info=True, opts=mbfir.make_opts(grid_m=m, lanes=8))
bad = 0
for q, (job, (h, st, i)) in enumerate(zip(jobs, res)):
    h1, s1, i1 = mbfir.fir_qp_cvx(*job[1], ctx=ctx, info=True, opts=mbfir.make_opts(grid_m=m))
    same = st == s1 and i["iters"] == i1["iters"] and i["dd_iters"] == i1["dd_iters"] and i["dd_kmax"] == i1["dd_kmax"] and i["pcost"] == i1["pcost"] and np.array_equal(h, h1)
    bad += not same
    print(q, "lanes", i["lanes"], st, s1, "iters", i["iters"], i1["iters"], "dd", i["dd_iters"], i1["dd_iters"], "kmax", i["dd_kmax"], i1["dd_kmax"],
          "pcost %.17g %.17g" % (i["pcost"], i1["pcost"]), "SAME" if same else "DIFFERENT", flush=True)
print("config 3 at size, unit of 8 against single solves: %d of 8 differ" % bad)
