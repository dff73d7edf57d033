import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from oracle import ddlin
for n in (3, 20, 40, 70):
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n + 30, n))
    Hw = B.T @ B; Hw = 0.5 * (Hw + Hw.T)
    b = rng.standard_normal((2, n)); bl = np.zeros((2, n))
    xh, xl, nfix, Lh, Ll = mbfir.test_ddsolve(Hw, np.zeros((0, n)), np.zeros(0), b, bl, factor=True)
    Hh, Hl = Hw.copy(), np.zeros_like(Hw)
    d0 = np.diag(Hh).copy()
    ddlin.chol(Hh, Hl, 1e-28, d0)
    Oh, Ol = np.tril(Hh), np.tril(Hl)
    dL = (Lh - Oh) + (Ll - Ol)
    Bh, Bl = np.ascontiguousarray(b.T), np.ascontiguousarray(bl.T)
    ddlin.cho_solve(Hh, Hl, Bh, Bl)
    dx = (xh - Bh.T) + (xl - Bl.T)
    # residual of L L' = H in longdouble
    Ld = Lh.astype(np.longdouble) + Ll.astype(np.longdouble)
    Od = Oh.astype(np.longdouble) + Ol.astype(np.longdouble)
    rd = np.abs(np.tril(Ld @ Ld.T - Hw)).max(); ro = np.abs(np.tril(Od @ Od.T - Hw)).max()
    print("n", n, "max|dL|", np.abs(dL).max(), "at", np.unravel_index(np.abs(dL).argmax(), dL.shape), "max|dx|", np.abs(dx).max(),
          "resid dev", float(rd), "orc", float(ro), "Ll absmax", np.abs(Ll).max(), flush=True)
    if n == 3:
        print(Lh, Ll, Oh, Ol, sep="\n")
