"""fir_ap_cvx(100, S-C13 spec of dzrf_mb(100, 0.04, ...), obj=1): device taps against the oracle's, pass-band |H|."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import numpy as np
import conftest  # noqa: F401
import mbfir
from oracle import designers
n, dt = 100, 0.04
cf = mbfir.spec.spectrum_c13(14.0)[[5, 0, 2, 3, 1]] * 1e-3; cf = cf - cf[4]
f, a, d = mbfir.spec.band_spec(n, dt, list(cf), [0.1] * 5, [0, 0, 0, 0, 60], [.005] * 4 + [.01], "ex")
w = np.linspace(f[8], f[9], 9) * np.pi
k = np.arange(n)
for obj in (1.0, 0.1):
    ho, so, io = designers.fir_ap_cvx(n, f, a, d, obj, 1e-3, info=True)
    for dense in (0, 1):
        hg, sg, ig = mbfir.fir_ap_cvx(n, f, a, d, obj, 1e-3, info=True, opts=mbfir.make_opts(dense_trig=dense))
        H = lambda h: np.abs((h[None, :] * np.exp(-1j * w[:, None] * k[None, :])).sum(1))
        print("obj %.1f dense %d: gpu %s it %d pcost %.10e | oracle %s it %d pcost %.10e | taps relinf %.2e" % (
            obj, dense, sg, ig["iters"], ig["pcost"], so, io["iters"], io["pcost"], np.abs(hg - ho).max() / np.abs(ho).max()))
        print("   |H| gpu   ", np.round(H(hg), 5))
        print("   |H| oracle", np.round(H(ho), 5), " allowed [%.5f, %.5f]" % (a[8] - d[4], a[8] + d[4]))
