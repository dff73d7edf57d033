"""STUDY (round 6): a centrality corrector on the second-order cones.  Oracle only: iteration counts with corrector=False and with
the adopted rule (conic_ipm.solve: the orthant rows and, on the plain path, the big cone -- programs with orthant rows only).  The
study that chose the rule ran a three-way flag in conic_ipm.solve (orthant rows only / + the 3-row cones / + all cones, on the
extended-precision path with one refinement pass) over these cases, 14 fir_ap_cvx instances and the H-1 dual-band family of
BASELINE config 3; its numbers are in DESIGN.md section 5a.  With the adopted rule fir_qp_cvx (no orthant rows) is unchanged and
fir_qprog_phs gains."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
import numpy as np
_sp = importlib.util.spec_from_file_location("mbfir_spec", os.path.join(ROOT, "multiband-rf-pulse-design_amd", "spec.py"))
spec = importlib.util.module_from_spec(_sp); _sp.loader.exec_module(spec)
from oracle import assemble, conic_ipm, designers
from conftest import CASES

def run(tag, P, dd):
    out = []
    for cc in (False, True):
        t = time.time()
        r = conic_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"], ddkkt=dict(theta=designers.DDKKT_THETA) if dd else None, corrector=cc)
        out.append("corrector=%s: status %d iters %d taken %d/%d pcost %.10e (%.1fs)" % (cc, r["status"], r["iters"], r["correctors_taken"], r["correctors"], r["pcost"], time.time() - t))
    print(tag, "l=%d nq3=%d big=%d" % (P["l"], P["nq3"], P["big"])); [print("    ", o) for o in out]; sys.stdout.flush()

for name in ("qp_modelB25", "qp_modelA48"):
    fn, args = CASES[name]
    n, f, a, d, k, obj = args
    P = assemble.assemble_fir_qp_cvx(n, f, a, d, k, obj, 0) if True else None
    run(name, P, True)
for name in ("qphs21", "qphs22"):
    fn, args = CASES[name]
    P = assemble.assemble_fir_qprog_phs(*args, 0) if hasattr(assemble, "assemble_fir_qprog_phs") else None
    if P is not None: run(name, P, False)
for n, m in ((256, 4096),) + (((384, 6144),) if len(sys.argv) > 1 else ()):
    f, a, d = spec.spec_h1_dualband(n)
    P = assemble.assemble_fir_qp_cvx(n, f, a, d, 120.0, 1e6, m)
    run("h1qp n=%d m=%d" % (n, m), P, True)
