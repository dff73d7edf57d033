"""A longer run of tests/test_fuzz_gpu.py's random specs than the suite holds (seeds lo..hi-1; the suite runs 0..63 and
0..11): device against the oracle, every failure listed instead of stopping at the first.
    python tools/gpu_fuzz_campaign.py 64 1064 [qp_lo qp_hi]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from oracle import conic_ipm, designers
import test_fuzz_gpu as F

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, illcond, t0 = [], [], time.time()
counts = {}
for seed in range(lo, hi):
    which, args = F.make_case(seed)
    try:
        hg, sg, ig = getattr(mbfir, which)(*args, info=True)
    except Exception as e:                                  # an error return of the device path
        bad.append((seed, which, "device raised %r" % (e,))); continue
    ho, so, io = getattr(designers, which)(*args, info=True)
    counts[(which, sg)] = counts.get((which, sg), 0) + 1
    if sg != so:
        bad.append((seed, which, "verdict device %s (rc %d) oracle %s (%s)" % (sg, ig["rc"], so, io["status"]))); continue
    if sg == "Solved":
        if abs(ig["pcost"] - io["pcost"]) > 1e-6 * max(1.0, abs(io["pcost"])):
            bad.append((seed, which, "objective %.12g vs %.12g" % (ig["pcost"], io["pcost"]))); continue
        clean = io["status"] == conic_ipm.STATUS_OPTIMAL and ig["relgap"] <= 1e-6
        if clean and np.max(np.abs(hg - ho)) > 1e-6 * max(np.max(np.abs(ho)), 1e-3):
            # how far apart are the CONIC solutions?  (fir_ap_cvx's taps are that solution seen through the spectral factorisation --
            # the log of a spectrum that may touch 1e-20, fir_ap_cvx.m:281,296 --, which amplifies by up to 1e9 on some instances)
            z = mbfir.get_context().last_solution(ig["n_unknowns"])
            xo = np.asarray(io["x"]); nn = min(len(z), len(xo))
            xd = float(np.abs(z[:nn] - xo[:nn]).max() / np.abs(xo[:nn]).max())
            (illcond if xd <= 1e-9 else bad).append((seed, which, "taps differ by %.3g relative to the largest (iters %d / %d), conic solutions %.1e apart" % (
                np.max(np.abs(hg - ho)) / max(np.max(np.abs(ho)), 1e-3), ig["iters"], io["iters"], xd)))
    if (seed - lo) % 50 == 49:
        print("seeds %d..%d done, %d failures so far, %.0f s" % (lo, seed, len(bad), time.time() - t0), flush=True)
print("random specs %d..%d: %d failures, %d more with taps apart at conic solutions equal to 1e-9 (conditioning of the spectral factorisation); verdict counts %s" % (
    lo, hi - 1, len(bad), len(illcond), sorted(counts.items())))
for b in bad:
    print("  FAIL", b)
for b in illcond:
    print("  ill-conditioned taps", b)

# ---- fir_qp_cvx the way dzrf_mb calls it (large k, obj): seeds qlo..qhi-1 of test_random_quadratic_phase_specs_with_large_peak_weights
if len(sys.argv) > 4:
    qlo, qhi = int(sys.argv[3]), int(sys.argv[4])
    bad, t0, nsolved, ndd = [], time.time(), 0, 0
    for seed in range(qlo, qhi):
        rng = np.random.default_rng(4000 + seed)
        n = int(rng.integers(40, 120))
        f, k = F.random_bands(rng, -1.0, 1.0, 3)
        f = f * rng.uniform(0.15, 0.5)
        amp = np.where(rng.random(k) < 0.4, 0.0, rng.uniform(0.5, 0.9, k))
        amp[int(rng.integers(0, k))] = 0.8
        a, d = np.repeat(amp, 2), rng.uniform(0.01, 0.05, k)
        kq, obj = float(rng.uniform(40, 140)), float(10 ** rng.uniform(4, 6.5))
        grid = int(rng.choice([0, 4 * n, 8 * n]))
        try:
            hg, sg, ig = mbfir.fir_qp_cvx(n, f, a, d, kq, obj, opts=mbfir.make_opts(grid_m=grid), info=True)
        except Exception as e:
            bad.append((seed, n, "device raised %r" % (e,))); continue
        ho, so, io = designers.fir_qp_cvx(n, f, a, d, kq, obj, grid_m=grid, info=True)
        nsolved += sg == "Solved"; ndd += ig["dd_iters"] > 0
        if sg != so:
            bad.append((seed, n, "verdict device %s (rc %d, relgap %.2g) oracle %s (%s)" % (sg, ig["rc"], ig["relgap"], so, io["status"]))); continue
        if sg == "Solved":
            if abs(ig["pcost"] - io["pcost"]) > 1e-6 * max(1.0, abs(io["pcost"])):
                bad.append((seed, n, "objective %.12g vs %.12g" % (ig["pcost"], io["pcost"]))); continue
            if io["status"] == conic_ipm.STATUS_OPTIMAL and ig["relgap"] <= 1e-8 and np.max(np.abs(hg - ho)) > 1e-4 * max(np.max(np.abs(ho)), 1e-3):
                bad.append((seed, n, "taps differ by %.3g (iters %d / %d)" % (np.max(np.abs(hg - ho)), ig["iters"], io["iters"])))
        if (seed - qlo) % 10 == 9:
            print("qp seeds %d..%d done, %d failures so far, %.0f s" % (qlo, seed, len(bad), time.time() - t0), flush=True)
    print("large-weight fir_qp_cvx specs %d..%d: %d failures; %d solved, %d used the extended-precision solve" % (qlo, qhi - 1, len(bad), nsolved, ndd))
    for b in bad:
        print("  FAIL", b)
