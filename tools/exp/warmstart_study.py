"""does a warm start of the homogeneous self-dual embedding (Skajaa, Andersen & Ye: convex combination of
a neighbouring problem's optimum with a cold point) remove >= 25 % of the IPM iterations on the sequences the reference's callers
produce?  Oracle only (oracle/conic_ipm.py, start=...).
  (a) the bench's Peak sweep (bSSFP_pulse_diff_Peak.m:68): S-C13, fixed ripples, Peak log-spaced -- neighbours differ in the
      right-hand side of the spike cones only (same rows);
  (b) a ripple sweep (the bench's second axis): neighbours differ in the band bounds (same rows);
  (c) the transition-width bisection of fir_ap.m:63-106: neighbours differ in the band edges -- the rows change (points move
      between band and transition, idx_stop changes), so s*, z* are carried over PER FREQUENCY (matched by the frequency value and
      the row's kind; new rows get the cold values).
    python tools/exp/warmstart_study.py [n]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
from conftest import c13
from oracle import assemble, conic_ipm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def solve(P, start=None):
    return conic_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"], start=start)


def run(label, progs, carry):
    """progs: list of assembled programs; carry(prev_prog, prev_result, prog) -> (x, s, z) for prog"""
    cold = [solve(P) for P in progs]
    print("%s: cold iterations %s, status %s" % (label, [r["iters"] for r in cold], sorted({r["status"] for r in cold})), flush=True)
    for kind in ("e", "own"):
        for lam in (0.5, 0.9, 0.99):
            its, bad = [], 0
            for q in range(1, len(progs)):
                x, s, z = carry(progs[q - 1], cold[q - 1], progs[q])
                r = solve(progs[q], start=dict(x=x, s=s, z=z, lam=lam, cold=kind))
                its.append(r["iters"])
                if r["status"] != cold[q]["status"] or abs(r["pcost"] - cold[q]["pcost"]) > 1e-6 * max(1, abs(cold[q]["pcost"])):
                    bad += 1
            base = sum(r["iters"] for r in cold[1:])
            print("   cold point %-3s lambda %.2f: iterations %s  total %d vs %d cold (%+.0f %%), %d verdict/objective changes"
                  % (kind, lam, its, sum(its), base, 100.0 * (sum(its) - base) / base, bad), flush=True)


same = lambda Pp, rp, P: (rp["x"], rp["s"], rp["z"])
f, a, d = c13(n, "duration")
peaks = np.logspace(-4, -2, 16)[4:10]
run("(a) Peak sweep, n=%d" % n, [assemble.assemble_fir_ap_cvx(n, f, a, d, 0.1, float(p)) for p in peaks], same)
run("(b) ripple sweep, n=%d" % n, [assemble.assemble_fir_ap_cvx(n, f, a, [x * 2 ** (j / 4) for x in d], 0.1, 1e-3) for j in range(5)], same)


def widen(f, fa):
    f = np.array(f, float).copy(); f[0::2] -= fa; f[1::2] += fa
    return list(f)


def carry_rows(Pp, rp, P):
    """s*, z* of the previous program mapped onto the rows of the new one: LP rows are matched by (frequency value, sign pattern
    of the row in G's first column block), everything that finds no partner gets e; cone rows (same count) are copied."""
    lp, l = Pp["l"], P["l"]
    s, z = np.ones(len(P["h"])), np.ones(len(P["h"]))
    s[l:], z[l:] = rp["s"][lp:], rp["z"][lp:]
    key = lambda Q, r: (round(float(Q["G"][r, 1] / (abs(Q["G"][r, 0]) + 1e-300)), 9), float(np.sign(Q["G"][r, 0])), float(Q["G"][r, -1]))
    prev = {}
    for r in range(lp):
        prev.setdefault(key(Pp, r), r)
    for r in range(l):
        q = prev.get(key(P, r))
        if q is not None:
            s[r], z[r] = max(rp["s"][q], 1e-300), max(rp["z"][q], 1e-300)
    return rp["x"], s, z


fadds = [0.0, 0.0020, 0.0010, 0.0015, 0.00125]            # a bisection's probes (feasible ones and infeasible ones mixed)
run("(c) transition-width probes, n=%d" % n, [assemble.assemble_fir_ap_cvx(n, widen(f, fa), a, d, 0.1, 1e-3) for fa in fadds], carry_rows)
