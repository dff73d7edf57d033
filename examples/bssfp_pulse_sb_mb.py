"""bSSFP_pulse_sb_mb.m on the MI355X: C-13 multiband excitation pulse (one metabolite excited, four left alone),
minimum-order arbitrary-phase design, inverse SLR, simulated profile, optional Varian file.

    python examples/bssfp_pulse_sb_mb.py [lactate|pyruvate|urea|alanine] [--min-order 58] [--write]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbfir  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("compound", nargs="?", default="lactate")
ap.add_argument("--min-order", type=float, default=58, help="<= 1: fraction of the bisection (fir_ap.m), > 1: fixed tap count")
ap.add_argument("--write", action="store_true", help="write <root>.RF (rfwrite_varian.m)")
args = ap.parse_args()

B0, n, T, FA, d1, d2 = 14.0, 100, 4.0, 60.0, 0.01, 0.005           # bSSFP_pulse_sb_mb.m:9-15
gamma = 1.0705
pick = [5, 0, 2, 3, 1]                                                # urea, pyruvate, alanine, pyruvate hydrate, lactate
sel = {"urea": 0, "pyruvate": 1, "alanine": 2, "lactate": 4}[args.compound]
cf = mbfir.spec.spectrum_c13(B0)[pick] * 1e-3                         # kHz
cf = cf - cf[sel]
mb_FA = [FA if i == sel else 0.0 for i in range(5)]
mb_ripple = [d1 if i == sel else d2 for i in range(5)]
dt = T / n
dt_epic = 4e-3                                                        # :68-73: snap to the scanner's 4 us raster
if abs(dt / dt_epic - round(dt / dt_epic)) > 1e-9:
    dt = dt_epic * np.floor(dt / dt_epic)
t0 = time.time()
rf, b, rf_spec, b_spec = mbfir.dzrf_mb(n, dt, list(cf), [0.1] * 5, mb_FA, mb_ripple, "ex", "ap_minorder_cvx", "C-13", 0, 1, None, 0,
                                       args.min_order, probes=4)
t_design = time.time() - t0
if len(rf) == 0:
    sys.exit("Filter design failed.")
print("mb-ap-SLR computation time: %.4f s" % t_design)
print("pulse duration: %.3f ms (%d samples)" % (len(rf) * dt, len(rf)))
print("pulse power: %.4f G^2*ms   peak amplitude: %.4f G" % (np.sum(np.abs(rf) ** 2) * dt, np.max(np.abs(rf))))
fs = 1 / dt
fk = np.linspace(-fs / 2, fs / 2, 2048)                               # sim_rf_spectral.m: 2048 off-resonances
a, bb = mbfir.abr(rf * (2 * np.pi * gamma * dt), fk * len(rf) * dt)
mxy = np.abs(2 * np.conj(a) * bb)
f = np.asarray(rf_spec["f"]) * fs / 2
for i, name in enumerate(["urea", "pyruvate", "alanine", "pyr. hydrate", "lactate"]):
    m = (fk >= f[2 * i]) & (fk <= f[2 * i + 1])
    print("  %-13s [%7.3f, %7.3f] kHz  |Mxy| in [%.4f, %.4f]   spec %.4f +- %.4f" % (
        name, f[2 * i], f[2 * i + 1], mxy[m].min(), mxy[m].max(), rf_spec["a"][2 * i], rf_spec["d"][i]))
if args.write:
    root = "mbslr_%s_%gdeg_%gms" % (args.compound, FA, len(rf) * dt)
    print("wrote", mbfir.rfwrite_varian(rf, len(rf) * dt, None, root))
