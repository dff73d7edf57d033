"""dzrf_mb over the filter types on the S-C13 spec: worst in-band deviation of the simulated |Mxy| per band."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import numpy as np
import conftest  # noqa: F401
import mbfir
cf = mbfir.spec.spectrum_c13(14.0)[[5, 0, 2, 3, 1]] * 1e-3; cf = list(cf - cf[4])
FA, rp = [0, 0, 0, 0, 60], [.005] * 4 + [.01]
dt, gamma = 0.04, 1.0705
for ftype, kw in (("ap_cvx", {}), ("ap_minstopripple_cvx", {}), ("ap_minorder_cvx", dict(probes=4)), ("ap_mintran_cvx", dict(probes=3))):
    rf_pulse, b, rf_spec, b_spec = mbfir.dzrf_mb(100, dt, cf, [0.1] * 5, FA, rp, "ex", ftype, "C-13", **kw)
    rf = rf_pulse * (2 * np.pi * gamma * dt)
    fs = 1 / dt
    fk = np.linspace(-fs / 2, fs / 2, 8001)[:-1]
    x = fk * len(rf) * dt
    out = []
    for hp in (True, False):
        a_, b_ = mbfir.abrm(rf, -x, hard_pulse=True) if hp else mbfir.abrm(rf, x)
        mxy = np.abs(2 * np.conj(a_) * b_)
        f = np.asarray(rf_spec["f"]) * fs / 2
        dev = []
        for i in range(5):
            sel = (fk >= f[2 * i]) & (fk <= f[2 * i + 1])
            dev.append(np.max(np.abs(mxy[sel] - rf_spec["a"][2 * i])) / rf_spec["d"][i])
        out.append(" ".join("%.3f" % v for v in dev))
    print("%-22s n=%3d  band widths (Hz) %s | dev/ripple hard-pulse: %s | abrm: %s" % (ftype, len(b), np.round(np.diff(np.asarray(rf_spec["f"]) * fs / 2)[0::2] * 1e3, 1), out[0], out[1]), flush=True)
