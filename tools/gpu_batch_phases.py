"""Where a headline batch's wall time goes on the host side: per unit, the time from the start of mbfir_solve_batch to the start
of the unit's solve (assembly of all jobs, grouping, waiting for a context), the solve itself and what follows it."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); warnings.filterwarnings("ignore")
import mbfir
from gpu_lanes import jobs_for
jobs = jobs_for(512, 64)
o = mbfir.make_opts(grid_m=16384, lanes=16)
mbfir.solve_batch(jobs, streams=4, opts=o)
# the C call alone, inside the Python wrapper's own time
lib = mbfir.load_library()
_c_call = lib.mbfir_solve_batch
c_ms = []
class _Timed:
    def __getattr__(self, name):
        if name != "mbfir_solve_batch":
            return getattr(lib, name)
        def f(*a):
            t = time.perf_counter(); r = _c_call(*a); c_ms.append((time.perf_counter() - t) * 1e3); return r
        return f
mbfir._lib = _Timed()
for _ in range(4):
    t = time.time()
    res = mbfir.solve_batch(jobs, streams=4, info=True, opts=o)
    dt = (time.time() - t) * 1e3
    units = {}
    for q, r in enumerate(res):
        i = r[2]
        units.setdefault((round(i["ms_assemble"], 3), round(i["ms_solve"], 3)), []).append((q, i["iters"], i["ms_post"], i["ms_total"]))
    print("batch %.1f ms, of which the C call %.1f ms" % (dt, c_ms[-1]))
    for (a, s), js in sorted(units.items()):
        print("  unit of %2d designs (jobs %d..%d): solve starts at %.1f ms, solve %.1f ms, iterations %d..%d, post %.1f ms, done at %.1f ms" % (
            len(js), js[0][0], js[-1][0], a, s, min(j[1] for j in js), max(j[1] for j in js), max(j[2] for j in js), max(j[3] for j in js)))
