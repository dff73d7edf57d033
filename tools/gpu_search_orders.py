"""The min-order search of fir_ap.m:143-176 on S-C13 (dt fixed: the band edges of n stay, the order moves; bSSFP_pulse_diff_Peak.m:72-77
calls it through dzrf_mb): the reference's bisection (probes = 1), four probes per round one design per stream (rounds 3-4), and the
same probes as lock-step units of different orders (round 5).  Prints wall-clock per search and the probes.
    python tools/gpu_search_orders.py [n = 100] [regime: minorder | duration]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from conftest import c13
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
f, a, d = c13(n, sys.argv[2] if len(sys.argv) > 2 else "duration")
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
o = mbfir.make_opts(grid_m=grid) if grid else None
for label, kw in (("bisection, probes=1", dict(probes=1)), ("probes=4, one design per stream", dict(probes=4, unit_probes=False)),
                  ("probes=4 as lock-step units", dict(probes=4, unit_probes=True)), ("probes=8 as lock-step units", dict(probes=8, unit_probes=True))):
    for rep in range(2):                                   # the first pass warms allocations of these shapes
        log = []
        t = time.time()
        h, status, n_op, f_op = mbfir.fir_ap(n, f, a, d, 1e-3, 1, 0, log=log, opts=o, **kw)
        dt = time.time() - t
    print("n %d %-34s %.3f s  %s  %d probes  n_op %d" % (n, label, dt, status, len(log), n_op), flush=True)
