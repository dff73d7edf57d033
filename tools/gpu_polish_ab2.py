"""The final approach's factor (MBFIR_POLISH_APPROACH) on one box: headline batch, config 4, and over 800 fuzz specs how many designs of
the designers WITHOUT a default extended-precision path end up retried there (the numerical wall in plain double)."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from gpu_lanes import jobs_for
from test_fuzz_gpu import make_case
from conftest import CASES
jobs = jobs_for(512, 64); o = mbfir.make_opts(grid_m=16384, lanes=16)
jobs4 = jobs_for(200, 256); o4 = mbfir.make_opts(grid_m=4096, lanes=32)
cases = [make_case(s) for s in range(0, 800)]
fq, aq, dq = CASES["qphs21"][1][1:4]
for tag, env in [("approach %s, +%s" % (a, k), {"MBFIR_POLISH_APPROACH": a, "MBFIR_POLISH_SWEEPS": k}) for a, k in (("1000", "2"), ("100", "2"), ("30", "2"), ("10", "2"), ("1", "2"), ("30", "1"), ("1", "0"))]:
    os.environ.update(env)
    mbfir.solve_batch(jobs, streams=4, opts=o)
    best = 1e9
    for rep in range(3):
        t = time.time(); res = mbfir.solve_batch(jobs, streams=4, info=True, opts=o); best = min(best, time.time() - t)
    its = [r[2]["iters"] for r in res]
    passes = sum(r[2]["gv_passes"] + r[2]["gtv_passes"] for r in res) / max(1, sum(its))
    mbfir.solve_batch(jobs4, streams=4, opts=o4)
    b4 = 1e9
    for rep in range(2):
        t = time.time(); res4 = mbfir.solve_batch(jobs4, streams=4, info=True, opts=o4); b4 = min(b4, time.time() - t)
    retried, fz_its, failed = 0, 0, 0
    for which, args in cases:
        h, s, i = getattr(mbfir, which)(*args, info=True)
        fz_its += i["iters"]
        if which != "fir_qp_cvx" and i["dd_iters"] > 0: retried += 1
    h, s, i29 = mbfir.fir_qprog_phs(29, fq, aq, dq, info=True)
    for k in env: os.environ.pop(k)
    print("%-22s headline %.1f designs/s, %.1f iterations, %.1f passes per iteration;  config 4: %.0f designs/s;  800 fuzz specs: %d iterations, %d retried in extended precision;  the 29-tap fir_qprog_phs: %d iterations (%d extended)" % (
        tag, 64 / best, np.mean(its), passes, 256 / b4, fz_its, retried, i29["iters"], i29["dd_iters"]), flush=True)
