"""Lock-step batch exploration: designs/s for (lanes per unit, streams) on config 4 and the headline size."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import numpy as np
import mbfir

def jobs_for(n, count):
    peaks = np.logspace(-4, -2, 16)
    jobs = []
    for q in range(count):
        j, p = divmod(q, 16)
        f, a, d = mbfir.spec.spec_c13_bssfp(n, d1=0.01 * 2 ** ((j % 16) / 4), d2=0.005 * 2 ** ((j % 16) / 4))
        jobs.append(("fir_ap_cvx", (n, f, a, d, 0.1, float(peaks[p]))))
    return jobs

def run(n, m, count, lanes, streams, ref=None):
    jobs = jobs_for(n, count)
    o = mbfir.make_opts(grid_m=m, lanes=lanes)
    mbfir.solve_batch(jobs[:max(streams, lanes if lanes > 0 else 1)], streams=streams, opts=o)      # warm-up (arena growth)
    t = time.time()
    res = mbfir.solve_batch(jobs, streams=streams, info=True, opts=o)
    dt = time.time() - t
    ok = sum(1 for r in res if r[1] == "Solved")
    its = [r[2]["iters"] for r in res]
    pc = np.array([r[2]["pcost"] for r in res])
    dev = "" if ref is None else " max|dpcost| %.1e" % np.abs(pc - ref).max()
    print("n %d m %d: %3d designs lanes %2d streams %d: %.3f s = %7.1f designs/s, solved %d, iters %d..%d lanes_used %d%s" % (
        n, m, count, lanes, streams, dt, count / dt, ok, min(its), max(its), res[0][2]["lanes"], dev), flush=True)
    return pc

which = (sys.argv[1] if len(sys.argv) > 1 else "c4") if __name__ == "__main__" else "none"
if which == "c4":
    ref = run(200, 4096, 64, 1, 4)
    for lanes, streams in ((4, 4), (8, 4), (16, 4), (16, 2), (32, 2), (32, 1), (64, 1)):
        run(200, 4096, 64, lanes, streams, ref)
    run(200, 4096, 256, 32, 4)
    run(200, 4096, 256, 32, 2)
    run(200, 4096, 256, 16, 4)
elif which == "head":
    ref = run(512, 16384, 16, 1, 4)
    for lanes, streams in ((2, 4), (4, 4), (4, 2), (8, 2), (8, 1), (16, 1)):
        run(512, 16384, 16, lanes, streams, ref)
    run(512, 16384, 32, 8, 4)
    run(512, 16384, 32, 16, 2)
