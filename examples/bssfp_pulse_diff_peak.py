"""bSSFP_pulse_diff_Peak.m on the MI355X: the same lactate pulse for several end-spike limits (`Peak`), designed as ONE
batch of independent designs (mbfir.solve_batch, one HIP stream each), then inverse SLR and the simulated stop bands.

    python examples/bssfp_pulse_diff_peak.py [n_designs]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbfir  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 3
peaks = [1e-2, 1e-3, 1e-4] if count == 3 else list(np.logspace(-2, -4, count))      # bSSFP_pulse_diff_Peak.m:66
n, T, gamma = 100, 4.0, 1.0705
dt = T / n
f, a, d = mbfir.spec.spec_c13_bssfp(n, T)
# minimum order for the tightest spike limit once (fir_ap.m bisection, 4 probes per round), then every Peak at that length
h0, s0, n_op, _ = mbfir.fir_ap(n, f, a, d, min(peaks), 1, 0, probes=4)
t0 = time.time()
res = mbfir.solve_batch([("fir_ap_cvx", (n_op, f, a, d, 0.1, float(p))) for p in peaks], streams=4)
t = time.time() - t0
print("%d designs of %d taps in %.3f s (%.1f designs/s)" % (len(peaks), n_op, t, len(peaks) / t))
fs = 1 / dt
fk = np.linspace(f[0] * fs / 2 - 0.5, f[-1] * fs / 2 + 0.5, 1024)
for p, (h, s) in zip(peaks, res):
    if s != "Solved":
        print("Spike limit %.4f: %s" % (p, s))
        continue
    rf = mbfir.rfscaleg(mbfir.b2rf(h[::-1]), len(h) * dt, gamma)
    al, be = mbfir.abr(rf * (2 * np.pi * gamma * dt), fk * len(rf) * dt)
    mxy = np.abs(2 * np.conj(al) * be)
    stop = np.zeros(len(fk), dtype=bool)
    for i in range(4):
        stop |= (fk >= f[2 * i] * fs / 2) & (fk <= f[2 * i + 1] * fs / 2)
    print("Spike limit %.4f: peak %.4f G, end sample %.5f G, max stop-band |Mxy| %.5f" % (p, np.abs(rf).max(), abs(rf[-1]), mxy[stop].max()))
