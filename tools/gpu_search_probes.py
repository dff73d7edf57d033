"""The transition-width search of fir_ap.m:63-106 on the H-1 dual-band spec at n=260 (what specsat_H1_dualband.m:50 runs through
dzrf_mb 'ap_mintran_cvx'): the reference's bisection (probes = 1), four speculative probes per round one design per stream
(round 3), and the same four probes as ONE heterogeneous lock-step unit (round 4).  Prints wall-clock per search and the probes."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
n = int(sys.argv[1]) if len(sys.argv) > 1 else 260
f, a, d = mbfir.spec.spec_h1_dualband(n)
for label, kw in (("bisection, probes=1", dict(probes=1)), ("probes=4, one design per stream", dict(probes=4, unit_probes=False)),
                  ("probes=4 as one lock-step unit", dict(probes=4, unit_probes=True)), ("probes=8 as one lock-step unit", dict(probes=8, unit_probes=True))):
    for rep in range(2):                                   # the first pass warms allocations of these shapes
        log = []
        t = time.time()
        h, status, n_op, f_op = mbfir.fir_ap(n, f, a, d, 1e-3, 0, 1.0, log=log, **kw)
        dt = time.time() - t
    print("%-34s %.3f s  %s  %d probes  f_add %.6f" % (label, dt, status, len(log), (f[1::2] - np.asarray(f_op)[1::2]).min() * -1), flush=True)
