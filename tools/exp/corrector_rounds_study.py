"""STUDY (round 6, oracle only): the centrality corrector in K rounds, other trial steps, other boxes -- on a scratch copy of
oracle/conic_ipm.py patched in memory (the repo's oracle is not touched).  Nine fir_ap_cvx instances (S-C13, 58 ... 200 taps); cost
model: an iteration 1, a corrector solve 0.12.  Results: DESIGN.md section 5a."""
import os, sys, types, warnings, importlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); warnings.filterwarnings("ignore")
import numpy as np
import oracle
from oracle import assemble
from conftest import c13, CASES
src = open("/root/repo/oracle/conic_ipm.py").read()
# ---- the corrector block as K rounds -------------------------------------------------------------
i0 = src.index("            at = min(1.0, alpha + CORR_DELTA)\n")
i1 = src.index("            ncorr[0] += 1\n            ncorr[1] += ncorr_ok\n")
block = '''            x2c, z2c, Gx2c = x2, z2, Gx2
            for kround in range(PROTO_K):
                at = min(1.0, max(alpha + CORR_DELTA, PROTO_MULT * alpha))
                mut = sigma * mu
                ll_ = cone.l
                v = (lam[:ll_] + at * dss[:ll_]) * (lam[:ll_] + at * wdz[:ll_])
                tt = np.minimum(np.maximum(v, CORR_BMIN * mut), CORR_BMAX * mut) - v
                tt = np.maximum(tt, -CORR_BMAX * mut)
                bzk = np.zeros(R)
                bzk[:ll_] = -Wm.wl * (tt / lam[:ll_])
                rk = kkt_solve(Wm, H, cf, np.zeros(N), bzk, plain=True)
                xk, zk, Gxk = rk[:3]
                nk = rk[3] if len(rk) > 3 else 0.0
                cand = direction(sigma, dk_c, x2c + xk, z2c + zk, Gx2c + Gxk)
                alpha_c = step_of(cand[5], cand[6], cand[3], cand[4], STEP)
                ncorr[0] += 1
                if nk <= max(REFTOL * nrm_c, CORR_ETA * float(np.linalg.norm(rx))) and alpha_c >= CORR_ACCEPT * alpha:
                    dx, ds, dz, dtau, dkap, dss, wdz = cand
                    alpha = alpha_c
                    x2c, z2c, Gx2c = x2c + xk, z2c + zk, Gx2c + Gxk
                    ncorr_ok = 1
                    ncorr[1] += 1
                    if alpha >= PROTO_GATE:
                        break
                else:
                    break
'''
src2 = src[:i0] + block + src[i1 + len("            ncorr[0] += 1\n            ncorr[1] += ncorr_ok\n"):]
src2 = src2.replace("MAX_SWEEPS = 8\n", "MAX_SWEEPS = 8\nPROTO_K = 1\nPROTO_MULT = 0.0\nPROTO_GATE = 2.0\n", 1)
mod = types.ModuleType("oracle.conic_proto"); mod.__package__ = "oracle"; mod.__file__ = "conic_proto.py"
exec(compile(src2, mod.__file__, "exec"), mod.__dict__)
cases = [("c13_58", CASES["ap_c13_58"][1]), ("c13_64", CASES["ap_c13_64"][1])]
for n, regime, peak in ((100, "duration", 1e-3), (128, "duration", 1e-3), (160, "duration", 1e-3), (200, "duration", 1e-4), (200, "duration", 1e-3), (200, "minorder", 1e-2), (160, "minorder", 1e-3)):
    cases.append(("c13%s_%d_%g" % (regime[0], n, peak), (n,) + tuple(c13(n, regime)) + (0.1, peak)))
Ps = [(nm, assemble.assemble_fir_ap_cvx(*a, 0)) for nm, a in cases]
def run(tag, **kw):
    for k, v in kw.items(): setattr(mod, k, v)
    its, solves = [], 0
    for nm, P in Ps:
        r = mod.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"])
        its.append(r["iters"]); solves += r["correctors"]
        assert r["status"] == 0, (nm, r["status"])
    print("%-44s iterations %s total %d, corrector solves %d -> cost ~ %.0f" % (tag, its, sum(its), solves, sum(its) * 1.0 + solves * 0.12), flush=True)
    for k in kw: setattr(mod, k, {"PROTO_K": 1, "PROTO_MULT": 0.0, "PROTO_GATE": 2.0, "CORR_DELTA": 0.5, "CORR_ACCEPT": 1.01, "CORR_BMIN": 0.1, "CORR_BMAX": 10.0}[k])
if len(sys.argv) > 1 and sys.argv[1] in ("sigma", "after"):
    run = lambda *a, **k: None
run("baseline (K=1, delta 0.5)")
run("delta 0.3", CORR_DELTA=0.3)
run("delta 0.2", CORR_DELTA=0.2)
run("delta 0.1, at >= 2 alpha", CORR_DELTA=0.1, PROTO_MULT=2.0)
run("delta 0.3, at >= 3 alpha", CORR_DELTA=0.3, PROTO_MULT=3.0)
run("K=2", PROTO_K=2)
run("K=3", PROTO_K=3)
run("K=2 while alpha < 0.5", PROTO_K=2, PROTO_GATE=0.5)
run("K=3 while alpha < 0.5", PROTO_K=3, PROTO_GATE=0.5)
run("K=3 while alpha < 0.3", PROTO_K=3, PROTO_GATE=0.3)
run("K=2, delta 0.3", PROTO_K=2, CORR_DELTA=0.3)
run("K=3, delta 0.3, while alpha<0.5", PROTO_K=3, CORR_DELTA=0.3, PROTO_GATE=0.5)
run("box [0.3, 3]", CORR_BMIN=0.3, CORR_BMAX=3.0)
run("box [0.03, 30]", CORR_BMIN=0.03, CORR_BMAX=30.0)
# ---- the cap on Mehrotra's sigma (SIGMA_MAX = 0.25 in the product) ---------------------------------------------------------------
_defaults = {"SIGMA_MAX": 0.25}
def run2(tag, **kw):
    for k, v in kw.items(): setattr(mod, k, v)
    its = []
    for nm, P in Ps:
        r = mod.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"])
        its.append(r["iters"] if r["status"] == 0 else -r["iters"])
    print("%-44s iterations %s total %d" % (tag, its, sum(abs(v) for v in its)), flush=True)
    for k in kw: setattr(mod, k, _defaults[k])
if len(sys.argv) > 1 and sys.argv[1] == "sigma":
    for sm in [float(v) for v in sys.argv[2:]] or (0.25, 0.4, 0.6, 0.8, 1.0):
        run2("SIGMA_MAX %.2f" % sm, SIGMA_MAX=sm)
# ---- after the cap on sigma was lowered (final build): the other constants once more ---------------------------------------------
if len(sys.argv) > 1 and sys.argv[1] == "after":
    _defaults.update({"STEP": 0.99, "CORR_DELTA": 0.5, "CORR_BMIN": 0.1, "CORR_BMAX": 10.0, "CORR_ACCEPT": 1.01, "PROTO_K": 1, "PROTO_GATE": 2.0, "SIGMA_MAX_CORR": 0.05})
    run2("final build")
    run2("STEP 0.995", STEP=0.995)
    run2("STEP 0.999", STEP=0.999)
    run2("STEP 0.98", STEP=0.98)
    run2("delta 0.3", CORR_DELTA=0.3)
    run2("delta 0.7", CORR_DELTA=0.7)
    run2("box [0.3, 3]", CORR_BMIN=0.3, CORR_BMAX=3.0)
    run2("box [0.03, 30]", CORR_BMIN=0.03, CORR_BMAX=30.0)
    run2("accept 1.0", CORR_ACCEPT=1.0)
    run2("accept 1.1", CORR_ACCEPT=1.1)
    run2("K=2", PROTO_K=2)
    run2("cap 0.02", SIGMA_MAX_CORR=0.02)
    run2("cap 0.1", SIGMA_MAX_CORR=0.1)
