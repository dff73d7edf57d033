"""Wall time of the device inverse SLR (host arrays in/out) beside the NumPy oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import conftest  # noqa: F401  (registers the package as `mbfir`)
import mbfir
from oracle import slr

for n in (64, 512, 1024, 2048):
    rng = np.random.default_rng(n)
    b = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.hanning(n)
    b *= 0.7 / np.max(np.abs(np.fft.fft(b, 16 * n)))
    mbfir.b2rf(b)
    t0 = time.perf_counter()
    for _ in range(5):
        rf = mbfir.b2rf(b)
    tg = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    ro = slr.b2rf(b)
    tc = time.perf_counter() - t0
    print("n=%4d  device %.2f ms   numpy oracle %.1f ms   max diff %.1e" % (n, tg * 1e3, tc * 1e3, np.max(np.abs(rf - ro))), flush=True)
