"""The headline batch and config 4's sweep under the late-round switches, on ONE box: the cap on sigma where the corrector runs
(MBFIR_SIGMA_MAX: 0.25 = before) and the extra sweeps of the final approach and the end game (MBFIR_POLISH_SWEEPS: 0 = none)."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from gpu_lanes import jobs_for
jobs = jobs_for(512, 64); o = mbfir.make_opts(grid_m=16384, lanes=16)
jobs4 = jobs_for(200, 256); o4 = mbfir.make_opts(grid_m=4096, lanes=32)
for tag, env in (("default (cap 0.05, +2 sweeps)", {}), ("cap 0.25, +2 sweeps", {"MBFIR_SIGMA_MAX": "0.25"}), ("cap 0.05, no extra sweeps", {"MBFIR_POLISH_SWEEPS": "0"}),
                 ("cap 0.05, +1 sweep", {"MBFIR_POLISH_SWEEPS": "1"}), ("cap 0.25, no extra sweeps", {"MBFIR_SIGMA_MAX": "0.25", "MBFIR_POLISH_SWEEPS": "0"}), ("default again", {})):
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
    for k in env: os.environ.pop(k)
    print("%-32s headline %.1f designs/s, %d solved, %.1f iterations (%d..%d), %.1f passes per iteration;  config 4: %.0f designs/s, %.1f iterations" % (
        tag, 64 / best, sum(1 for r in res if r[1] == "Solved"), np.mean(its), min(its), max(its), passes, 256 / b4, np.mean([r[2]["iters"] for r in res4])), flush=True)
