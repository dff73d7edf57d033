"""When does each lock-step unit of the bench's batch end?  64 headline designs, units of `lanes` on `streams` streams: per unit its
iteration range, its solve time (all units start together) and its factorisation time."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); warnings.filterwarnings("ignore")
import mbfir
from gpu_lanes import jobs_for
n, m, count, lanes, streams = [int(v) for v in sys.argv[1:6]]
jobs = jobs_for(n, count)
o = mbfir.make_opts(grid_m=m, lanes=lanes)
mbfir.solve_batch(jobs, streams=streams, opts=o)
for rep in range(2):
    t = time.time()
    res = mbfir.solve_batch(jobs, streams=streams, info=True, opts=o)
    dt = time.time() - t
    units = {}
    for q, r in enumerate(res):
        i = r[2]
        units.setdefault((round(i["ms_solve"], 3), round(i["ms_chol"], 3)), []).append((q, i["iters"]))
    print("%d designs, lanes %d, streams %d: %.1f ms = %.1f designs/s" % (count, lanes, streams, 1e3 * dt, count / dt))
    for (ms, mc), v in sorted(units.items()):
        its = [x[1] for x in v]
        print("   unit of %2d  iterations %3d..%3d (sum %4d)  assemble+solve ends at %6.1f ms  factorisations %5.1f ms  designs %s" % (
            len(v), min(its), max(its), sum(its), ms + res[v[0][0]][2]["ms_assemble"], mc, [x[0] for x in v][:4]))
