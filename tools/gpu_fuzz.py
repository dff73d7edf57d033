"""Robustness sweep on the GPU: random band specs through all four designers, lattice path against the dense
path (same verdict, same taps), plus batch-of-everything against the single calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mbfir
from conftest import relinf

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 12345)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60


def random_bands(lo, hi, kmax):
    k = int(rng.integers(2, kmax + 1))
    widths = rng.uniform(0.02, 0.12, k) * (hi - lo) / 2
    gaps = rng.uniform(0.04, 0.2, k + 1) * (hi - lo) / 2
    tot = widths.sum() + gaps.sum()
    scale = (hi - lo) / tot
    edges, x = [], lo + gaps[0] * scale
    for w, g in zip(widths, gaps[1:]):
        edges += [x, x + w * scale]
        x += (w + g) * scale
    return np.array(edges), k


bad = 0
t0 = time.time()
for case in range(ncase):
    which = ["fir_ap_cvx", "fir_qp_cvx", "fir_linprog", "fir_qprog_phs"][case % 4]
    n = int(rng.integers(int(sys.argv[3]) if len(sys.argv) > 3 else 24, int(sys.argv[4]) if len(sys.argv) > 4 else 140))
    if which == "fir_linprog" and rng.random() < 0.5:
        f, k = random_bands(0.0, 1.0, 4)            # real filter
    else:
        f, k = random_bands(-1.0, 1.0, 5)
    amp = np.where(rng.random(k) < 0.5, 0.0, rng.uniform(0.3, 1.0, k))
    if not np.any(amp > 0): amp[int(rng.integers(0, k))] = 0.8
    a = np.repeat(amp, 2)
    d = rng.uniform(0.004, 0.03, k)
    if which == "fir_ap_cvx": args = (n, f, a, d, 0.1, 10 ** rng.uniform(-3, -1))
    elif which == "fir_qp_cvx": args = (n, f, a, d, float(rng.uniform(5, 60)), 1e3)
    elif which == "fir_linprog": args = (n, f, a, d)
    else: args = (n, f, a * np.exp(1j * rng.uniform(-0.3, 0.3, 2 * k).round(1).repeat(1)), d.astype(complex))
    if which == "fir_qprog_phs":                    # phase constant inside a band
        ph = np.repeat(np.exp(1j * rng.uniform(-0.3, 0.3, k)), 2)
        args = (n, f, a * ph, np.where(amp > 0, d * np.exp(0.3j), d.astype(complex)))   # pass bands need a phase ripple
    res = []
    for dense in (0, 1):
        try:
            h, s, i = getattr(mbfir, which)(*args, info=True, opts=mbfir.make_opts(dense_trig=dense))
            res.append((h, s, i["iters"], i["lattice"], i["rc"]))
        except Exception as e:                      # noqa: BLE001
            res.append((np.zeros(0), "EXC " + type(e).__name__ + ": " + str(e)[:60], -1, -1, -99))
    (hl, sl, il, ll, rl), (hd, sd, idn, ld, rd) = res
    diff = relinf(hl, hd) if sl == sd == "Solved" else 0.0
    flag = ""
    if sl != sd or diff > 1e-6 or (ll != 1 and not sl.startswith("EXC")):
        flag = "  <<<<<< MISMATCH"
        bad += 1
    print("%3d %-14s n=%3d k=%d  lattice: %-7s it %3d rc %2d | dense: %-7s it %3d rc %2d | taps diff %.1e%s" % (case, which, n, k, sl[:7], il, rl, sd[:7], idn, rd, diff, flag), flush=True)
print("cases %d mismatches %d  (%.1f s)" % (ncase, bad, time.time() - t0))
