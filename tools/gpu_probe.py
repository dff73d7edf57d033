"""Developer probe (GPU box): kernel hooks vs numpy, then one fir_ap_cvx solve next to the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbfir
from oracle import assemble, conic_ipm, specfact

rng = np.random.default_rng(0)
ctx = mbfir.get_context()
print("version", mbfir.load_library().mbfir_version())
print("fp64 peak TF (mfma, valu):", mbfir.mfma_peak())

# gram
for (m, nt, nw) in [(100, 37, 1), (1000, 200, 3), (5000, 399, 1)]:
    A = rng.standard_normal((m, nt)); d = rng.random((nw, m)) + 0.1
    T = mbfir.test_gram(A, d)
    ref = np.stack([(A.T * d[w]) @ A for w in range(nw)])
    print("gram", m, nt, nw, "relerr", np.abs(T - ref).max() / np.abs(ref).max())
# chol
for n in (50, 64, 200, 449):
    B = rng.standard_normal((n + 20, n)); H = B.T @ B + 0.1 * np.eye(n)
    L, M = mbfir.test_chol(H)
    Lr = np.linalg.cholesky(H)
    print("chol", n, "L err", np.abs(L - Lr).max() / np.abs(Lr).max(), "M err", np.abs(M @ Lr - np.eye(n)).max())
# specfact
for n in (16, 58, 100):
    h0 = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    r = np.correlate(h0, h0, mode="full")          # two-sided autocorrelation, length 2n-1
    rr = r[n - 1:]
    x = np.concatenate([[rr[0].real], rr[1:].real, rr[1:].imag])
    hg = mbfir.test_specfact(x, n)
    ho = specfact.fmp2(specfact.x_to_r(x, n))
    print("specfact", n, "relerr", np.abs(hg - ho).max() / np.abs(ho).max())

f100 = [-0.241994, -0.233994, -0.152431, -0.144431, -0.083851, -0.075851, -0.052798, -0.044798, -0.004, 0.004]
a = [0] * 8 + [0.500125, 0.500125]
d = [0.00250001] * 4 + [0.00866503]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = assemble.assemble_fir_ap_cvx(n, f100, a, d, 0.1, 1e-3)
hist = []
t0 = time.time()
r = conic_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"], history=hist)
print("oracle status", r["status"], "iters", r["iters"], "pcost %.12e" % r["pcost"], "time", time.time() - t0)
for i, hh in enumerate(hist[:6] + hist[-3:]):
    print("  o", hh["iters"], "%.10e %.10e gap %.2e pres %.1e dres %.1e" % (hh["pcost"], hh["dcost"], hh["gap"], hh["pres"], hh["dres"]))
ho = specfact.fmp2(specfact.x_to_r(r["x"][:2 * n - 1], n))
t0 = time.time()
hg, status, info = mbfir.fir_ap_cvx(n, f100, a, d, 0.1, 1e-3, dbg=1, info=True)
print("gpu", status, info["iters"], "pcost %.12e" % info["pcost"], "time", time.time() - t0)
print({k: info[k] for k in ("ms_assemble", "ms_solve", "ms_post", "ms_gram", "ms_chol")})
if status == "Solved":
    print("tap rel linf diff", np.abs(hg - ho).max() / np.abs(ho).max())
