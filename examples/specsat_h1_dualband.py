"""specsat_H1_dualband.m on the MI355X: dual-band H-1 spectral saturation pulse at 3 T (1.8-2.5 ppm at 120 degrees,
water at 90 degrees, 3-4.1 ppm untouched), `ap_mintran_cvx` design with the frequency axis centred during the design
(shift_f = 1), inverse SLR, simulated Mz, optional GE files.

    python examples/specsat_h1_dualband.py [--n 260] [--write]
"""
import argparse
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbfir  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=260)
ap.add_argument("--write", action="store_true", help="write specsat_h1_dualband.dat/.rho/.pha (rfwrite.m)")
args = ap.parse_args()
n, T, d1, d2, gamma = args.n, 26.0, 0.05, 0.001, 4.2576                 # specsat_H1_dualband.m:4-13
B0 = 127794577 / (42.577e6)
bands = [(1.8, 2.5), (3.0, 4.1), (4.8, 5.4)]                            # ppm; the last one is water
ref = sum(bands[2]) / 2
mb_cf = [[(lo - ref) * B0 * 42.577e-3, (hi - ref) * B0 * 42.577e-3] for lo, hi in bands]   # kHz
mb_FA, mb_range, mb_ripple = [120, 0, 90], [0.01] * 3, [d1, d2, d1]
dt = T / n
if abs(dt / 4e-3 - round(dt / 4e-3)) > 1e-9:                            # :38-43
    dt = 4e-3 * math.floor(dt / 4e-3)
    T = n * dt
t0 = time.time()
rf, b, rf_spec, b_spec = mbfir.dzrf_mb(n, dt, mb_cf, mb_range, mb_FA, mb_ripple, "sat", "ap_mintran_cvx", "H-1", 0, 1, None, 0,
                                       None, 0.95, 1, probes=3)
print("computation time: %.4f s" % (time.time() - t0))
if len(rf) == 0:
    sys.exit("Filter design failed.")
print("pulse duration:   %.3f ms" % (len(rf) * dt))
print("total power:      %.4f G^2*ms" % (np.sum(np.abs(rf) ** 2) * dt))
print("peak amplitude:   %.4f G" % np.max(np.abs(rf)))
fs = 1 / dt
fk = np.linspace(-fs / 2, fs / 2, 2048)
al, be = mbfir.abr(rf * (2 * np.pi * gamma * dt), fk * len(rf) * dt)
mz = 1 - 2 * np.abs(be) ** 2                                            # abr.m:12
f = np.asarray(rf_spec["f"]) * fs / 2
for i in range(3):
    m = (fk >= f[2 * i]) & (fk <= f[2 * i + 1])
    print("  band %d [%7.3f, %7.3f] kHz: Mz in [%.4f, %.4f]   spec %.4f +- %.4f" % (i, f[2 * i], f[2 * i + 1], mz[m].min(), mz[m].max(),
                                                                                 rf_spec["a"][2 * i], rf_spec["d"][i]))
if args.write:
    print("wrote", mbfir.rfwrite(rf, len(rf) * dt * 1e-3, mb_FA[2] * math.pi / 180, gamma * 1e3, 0, None, None, "specsat_h1_dualband"))
