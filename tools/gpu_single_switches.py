"""Single-design latency at given sizes under the host-side switches: default (speculative head), MBFIR_GRAPH=1 (the iteration as a
captured graph), MBFIR_SPECULATE=0 (neither)."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); warnings.filterwarnings("ignore")
import mbfir
from gpu_lanes import jobs_for
for n, m in ((512, 16384), (200, 4096), (64, 1024)):
    job = jobs_for(n, 1)[0]
    o = mbfir.make_opts(grid_m=m)
    for tag, env in (("default", {}), ("MBFIR_GRAPH=1", {"MBFIR_GRAPH": "1"}), ("MBFIR_SPECULATE=0", {"MBFIR_SPECULATE": "0"})):
        os.environ.update(env)
        ts = []
        for rep in range(6):
            t = time.time(); h, s, i = getattr(mbfir, job[0])(*job[1], opts=o, info=True); ts.append(time.time() - t)
        for k in env: os.environ.pop(k)
        print("n=%d m=%d %-18s %s %d iterations: best %.2f ms, median %.2f ms (chol %.2f ms)" % (n, m, tag, s, i["iters"], 1e3 * min(ts), 1e3 * sorted(ts)[3], i["ms_chol"]), flush=True)
