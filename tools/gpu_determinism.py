"""Run-to-run determinism: the bench's batch (64 headline designs, units of 16 on 4 streams), the heterogeneous batch and config 3 in
lock-step units, each solved `reps` times; every repetition must reproduce the first bit for bit (verdicts, iterations, objective, taps)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from gpu_lanes import jobs_for
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
def run(name, jobs, opts, streams):
    first, bad = None, 0
    for r in range(reps):
        res = mbfir.solve_batch(jobs, streams=streams, info=True, opts=opts)
        sig = [(s, i["iters"], i["pcost"], None if h is None else np.asarray(h).tobytes()) for h, s, i in res]
        if first is None: first = sig
        else: bad += sum(1 for a, b in zip(first, sig) if a != b)
    print("%-40s %d designs x %d repetitions: %d results differ from the first run" % (name, len(jobs), reps, bad), flush=True)
run("headline batch (units of 16, 4 streams)", jobs_for(512, 64), mbfir.make_opts(grid_m=16384, lanes=16), 4)
f, a, d = mbfir.spec.spec_h1_dualband(512)
run("config 3 (units of 4, 4 streams)", [("fir_qp_cvx", (512, f, a, [x * (1.0 + 0.02 * q) for x in d], 120.0, 1e6)) for q in range(16)], mbfir.make_opts(grid_m=16384), 4)
jobs = []
for seed in range(24):
    fr, ar, dr = mbfir.spec.spec_rand(128, seed)
    jobs.append(("fir_ap_cvx", (128, list(fr), list(ar * 0.5), list(dr), 0.1, 1e-2)))
run("heterogeneous (n = 128, units of 6, 3 streams)", jobs, mbfir.make_opts(lanes=6), 3)
