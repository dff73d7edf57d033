"""BASELINE config 3 as written (H-1 dual band, fir_qp_cvx k=120 obj=1e6, n=512, 16384 grid points; dzrf_mb.m:210-213)
as a batch: `count` designs (Peak-free program: the transition-band bound / ripple scaled per design so that they are
distinct), extended-precision KKT solve, one design per stream, `streams` in flight.  Prints designs/s."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
count, streams = int(sys.argv[1]), int(sys.argv[2])
n, m = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (512, 16384)
f, a, d = mbfir.spec.spec_h1_dualband(n)
jobs = [("fir_qp_cvx", (n, f, a, [x * (1.0 + 0.02 * q) for x in d], 120.0, 1e6)) for q in range(count)]
lanes = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # 0: the library's choice; 1: one design per stream (rounds 2-4)
o = mbfir.make_opts(grid_m=m, lanes=lanes) if lanes else mbfir.make_opts(grid_m=m)
mbfir.solve_batch(jobs[:max(streams, lanes * streams)], streams=streams, opts=o)
for _ in range(2):
    t = time.time()
    res = mbfir.solve_batch(jobs, streams=streams, info=True, opts=o)
    dt = time.time() - t
    print("config 3 as written, n %d m %d: %d designs, %d streams, lanes %s: %.3f s = %.2f designs/s; status %s iters %s dd_iters %s" % (
        n, m, count, streams, sorted(set(r[2]["lanes"] for r in res)), dt, count / dt, sorted(set(r[1] for r in res)), [r[2]["iters"] for r in res], [r[2]["dd_iters"] for r in res]), flush=True)
t = time.time()
h, st, inf = mbfir.fir_qp_cvx(*jobs[0][1], opts=o, info=True)
print("config 3 as written, one design alone: %.3f s, %s, %d iterations (%d extended-precision)" % (time.time() - t, st, inf["iters"], inf["dd_iters"]), flush=True)
