import sys, os, time, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import schur_kkt_proto as schur3
from oracle import assemble, conic_ipm
rng = np.random.default_rng(int(sys.argv[1]))
ncase = int(sys.argv[2])
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
chg = 0
for case in range(ncase):
    n = int(rng.integers(24, 140))
    f, k = random_bands(-1.0, 1.0, 5)
    amp = np.where(rng.random(k) < 0.5, 0.0, rng.uniform(0.3, 1.0, k))
    if not np.any(amp > 0): amp[int(rng.integers(0, k))] = 0.8
    a = np.repeat(amp, 2)
    d = rng.uniform(0.004, 0.03, k)
    objs = [1e3, 1e6, [0.1, 5.0], 10.0][case % 4]
    P = assemble.assemble_fir_qp_cvx(n, f, a, d, float(rng.uniform(5, 60)), objs)
    r0 = conic_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"])
    r1 = schur3.mod.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"])
    ok = lambda s: s in (0, 5)
    dx = np.abs(r0["x"] - r1["x"]).max() / max(np.abs(r0["x"]).max(), 1e-300) if ok(r0["status"]) and ok(r1["status"]) else 0
    flag = "" if r0["status"] == r1["status"] else ("  << status" + (" VERDICT" if ok(r0["status"]) != ok(r1["status"]) else ""))
    if flag: chg += 1
    print("%3d n=%3d k=%d obj %-10s base st %d it %3d fixes %4d relgap %.0e | schur st %d it %3d fixes %4d relgap %.0e | dx %.1e%s" % (case, n, k, objs, r0["status"], r0["iters"], r0["chol_fixes"], r0["relgap"], r1["status"], r1["iters"], r1["chol_fixes"], r1["relgap"], dx, flag), flush=True)
print("changed", chg)
