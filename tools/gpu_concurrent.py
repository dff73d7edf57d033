"""Throughput with K independent designs in flight on one GPU (one context + stream per host thread)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mbfir
from conftest import c13
n, m = 512, 16384
f, a, d = c13(n, "duration")
opts = mbfir.make_opts(grid_m=m)
for K in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    ctxs = [mbfir.Context(0) for _ in range(K)]
    def work(c, reps, out):
        for _ in range(reps):
            h, s, i = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=opts, ctx=c, info=True)
            assert s == "Solved"
        out.append(i["iters"])
    for reps in (1, 3):        # warm-up, then timed
        out = []
        ts = [threading.Thread(target=work, args=(c, reps, out)) for c in ctxs]
        t0 = time.time()
        for t in ts: t.start()
        for t in ts: t.join()
        el = time.time() - t0
    print("K=%d: %d designs in %.3f s -> %.2f designs/s (%.1f ms per design per stream)" % (K, K * reps, el, K * reps / el, el / reps * 1e3), flush=True)
    for c in ctxs: c.close()
