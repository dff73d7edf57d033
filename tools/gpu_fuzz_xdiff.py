"""Fuzz seeds whose taps differ from the oracle's: how far apart are the CONIC solutions (what the iteration computes; the taps of
fir_ap_cvx are that solution seen through the spectral factorisation, fir_ap_cvx.m:264-304)?"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from oracle import designers
from test_fuzz_gpu import make_case
for seed in [int(v) for v in sys.argv[1:]]:
    which, args = make_case(seed)
    ho, so, io = getattr(designers, which)(*args, info=True)
    hg, sg, ig = getattr(mbfir, which)(*args, info=True)
    z = mbfir.get_context().last_solution(ig["n_unknowns"])
    xo = np.asarray(io["x"])
    n = min(len(z), len(xo))
    print("seed %d %s n=%d: iterations %d / %d, objective %.3e apart (relative), conic solution %.2e apart (relative to its largest entry), taps %.2e; gap %.1e / %.1e" % (
        seed, which, args[0], ig["iters"], io["iters"], abs(ig["pcost"] - io["pcost"]) / max(1e-300, abs(io["pcost"])),
        np.abs(z[:n] - xo[:n]).max() / np.abs(xo[:n]).max(), np.abs(hg - ho).max() / max(np.abs(ho).max(), 1e-3), ig["gap"], io["gap"]), flush=True)
