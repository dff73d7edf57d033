"""Developer probe (GPU box): all four designers next to the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbfir
from oracle import designers

f100 = [-0.241994, -0.233994, -0.152431, -0.144431, -0.083851, -0.075851, -0.052798, -0.044798, -0.004, 0.004]
a = [0] * 8 + [0.500125, 0.500125]
d = [0.00250001] * 4 + [0.00866503]
fq = [-0.5, -0.3, -0.1, 0.1, 0.3, 0.5]; aq = [0, 0, 1, 1, 0, 0]; dq = [0.01, 0.02, 0.01]
fl = [0, 0.2, 0.3, 1]; al_ = [1, 1, 0, 0]; dl = [0.01, 0.01]
fl2 = [-1, -0.4, -0.2, 0.3, 0.5, 1]; al2 = [0, 0, 1, 0.8, 0, 0]; dl2 = [0.01, 0.02, 0.01]
fp = [-0.6, -0.3, -0.1, 0.1, 0.3, 0.6]; ac = np.array([0, 0, 1, 1, 0, 0], dtype=complex)
dc = np.array([0.02, 0.05 * np.exp(1j * 0.3), 0.02])
verbose = int(os.environ.get("V", "0"))
cases = [
    ("ap58", "fir_ap_cvx", (58, f100, a, d, 0.1, 1e-3)),
    ("ap50inf", "fir_ap_cvx", (50, f100, a, d, 0.1, 1e-3)),
    ("ap64", "fir_ap_cvx", (64, f100, a, d, 0.1, 1e-3)),
    ("qpB25", "fir_qp_cvx", (25, fq, aq, dq, 20.0, [0.1, 5.0])),
    ("qpA24inf", "fir_qp_cvx", (24, fq, aq, dq, 20.0, 100.0)),
    ("qpA48", "fir_qp_cvx", (48, fq, [0, 0, 1, 1, 0, 0], [0.05, 0.05, 0.05], 5.0, 10.0)),
    ("lin64", "fir_linprog", (64, fl, al_, dl)),
    ("lin31c", "fir_linprog", (31, fl2, al2, dl2)),
    ("lin32c", "fir_linprog", (32, fl2, al2, dl2)),
    ("lin31inf", "fir_linprog", (31, fl, al_, dl)),
    ("qphs21", "fir_qprog_phs", (21, fp, ac, dc)),
    ("qphs22", "fir_qprog_phs", (22, fp, ac, dc)),
]
only = sys.argv[1:] 
for name, fn, args in cases:
    if only and name not in only: continue
    t0 = time.time()
    ho, so, io = getattr(designers, fn)(*args, info=True)
    t1 = time.time()
    hg, sg, ig = getattr(mbfir, fn)(*args, dbg=verbose, info=True)
    t2 = time.time()
    line = "%-9s oracle %s it %d pcost %.10e (%.2fs) | gpu %s it %d pcost %.10e (%.3fs)" % (
        name, so, io["iters"], io["pcost"], t1 - t0, sg, ig["iters"], ig["pcost"], t2 - t1)
    if so == "Solved" and sg == "Solved":
        line += " | tap relerr %.2e" % (np.abs(hg - ho).max() / np.abs(ho).max())
    print(line, flush=True)
