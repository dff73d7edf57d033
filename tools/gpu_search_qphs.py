"""ss/fir_min_order_qprog_phs.m:95-120 on the device: the reference's bisection (probes = 1), four probes per round one design per
stream (opts.lanes = 1: what round 5 did for this designer -- its centred delays move the lattice origin with the order) and the
same probes as lock-step units (round 6: the origin is a per-lane dimension).  Prints wall-clock per search and the probe count."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
nmax = int(sys.argv[1]) if len(sys.argv) > 1 else 160
f, a, d = [-0.6, -0.3, -0.1, 0.1, 0.3, 0.6], [0, 0, 1, 1, 0, 0], [0.002, 0.01 * np.exp(0.3j), 0.002]
ref = None
for label, kw in (("bisection, probes=1", dict(probes=1)), ("probes=4, one design per stream", dict(probes=4, opts=mbfir.make_opts(lanes=1))),
                  ("probes=4 in lock-step units", dict(probes=4)), ("probes=8 in lock-step units", dict(probes=8))):
    for rep in range(2):                                   # the first pass warms allocations of these shapes
        log = []
        t = time.time()
        h, status = mbfir.fir_min_order_qprog_phs(nmax, f, a, d, log=log, **kw)
        dt = time.time() - t
    if ref is None:
        ref = h
    print("%-34s %.3f s  %s  %d taps  %d probes  taps vs bisection %.1e" % (label, dt, status, len(h), len(log),
          np.abs(h - ref).max() / np.abs(ref).max() if len(h) == len(ref) else -1), flush=True)
