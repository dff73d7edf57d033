"""Centrality corrector on / off (MBFIR_CORRECTOR): parity with the oracle on small cases, iterations and time of the bench's batch."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from oracle import designers
from gpu_lanes import jobs_for

def relinf(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())

what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("all", "parity"):
    ctx = mbfir.Context(0)
    f = [-0.6, -0.35, -0.1, 0.15, 0.45, 0.8]; a = [0, 0, 0.7, 0.7, 0, 0]; d = [0.01, 0.02, 0.01]
    for n, peak in ((33, 1e-2), (48, 1e-3), (21, 1e-1)):
        ho, so, io = designers.fir_ap_cvx(n, f, a, d, 0.1, peak, info=True)
        for corr in ("1", "0"):
            os.environ["MBFIR_CORRECTOR"] = corr
            hg, sg, ig = mbfir.fir_ap_cvx(n, f, a, d, 0.1, peak, ctx=ctx, info=True)
            print("fir_ap_cvx n %d peak %g corrector %s: device %s %d its (correctors %d, taken %d, G v %d, G'v %d)  oracle %s %d its (taken %d of %d)  taps %.2e" % (
                n, peak, corr, sg, ig["iters"], ig["correctors"], ig["correctors_taken"], ig["gv_passes"], ig["gtv_passes"], so, io["iters"], io["correctors_taken"], io["correctors"],
                relinf(hg, ho) if sg == so == "Solved" else -1), flush=True)
    fl = [0, 0.2, 0.3, 1]; al = [1, 1, 0, 0]; dl = [0.01, 0.01]
    ho, so, io = designers.fir_linprog(64, fl, al, dl, info=True)
    for corr in ("1", "0"):
        os.environ["MBFIR_CORRECTOR"] = corr
        hg, sg, ig = mbfir.fir_linprog(64, fl, al, dl, ctx=ctx, info=True)
        print("fir_linprog 64 corrector %s: device %s %d its (taken %d)  oracle %s %d its  taps %.2e" % (corr, sg, ig["iters"], ig["correctors_taken"], so, io["iters"], relinf(hg, ho)), flush=True)
    ctx.close()
if what in ("all", "batch"):
    n, m, count, lanes, streams = 512, 16384, 64, 16, 4
    jobs = jobs_for(n, count)
    o = mbfir.make_opts(grid_m=m, lanes=lanes)
    for corr in ("1", "0", "1", "0"):
        os.environ["MBFIR_CORRECTOR"] = corr
        mbfir.solve_batch(jobs[:16], streams=streams, opts=o)
        best = 1e9
        for rep in range(3):
            t = time.time(); res = mbfir.solve_batch(jobs, streams=streams, info=True, opts=o); best = min(best, time.time() - t)
        its = [r[2]["iters"] for r in res]
        print("batch of %d, corrector %s: %.1f ms = %.1f designs/s; solved %d; iterations mean %.1f (%d..%d); correctors taken %.0f %%; G v / G'v passes per iteration %.2f / %.2f" % (
            count, corr, 1e3 * best, count / best, sum(1 for r in res if r[1] == "Solved"), np.mean(its), min(its), max(its),
            100.0 * sum(r[2]["correctors_taken"] for r in res) / max(1, sum(r[2]["correctors"] for r in res)),
            np.mean([r[2]["gv_passes"] / max(r[2]["iters"], 1) for r in res]), np.mean([r[2]["gtv_passes"] / max(r[2]["iters"], 1) for r in res])), flush=True)
    one = jobs[0]
    ctx = mbfir.Context(0)
    for corr in ("1", "0"):
        os.environ["MBFIR_CORRECTOR"] = corr
        mbfir.fir_ap_cvx(*one[1], ctx=ctx, opts=mbfir.make_opts(grid_m=m))
        t = time.time(); h, s, i = mbfir.fir_ap_cvx(*one[1], ctx=ctx, info=True, opts=mbfir.make_opts(grid_m=m)); dt = time.time() - t
        print("one design, corrector %s: %.1f ms, %d iterations, %s" % (corr, 1e3 * dt, i["iters"], s), flush=True)
    ctx.close()
