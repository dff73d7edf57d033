"""BASELINE config 3 (fir_qp_cvx, H-1 dual band) one design: iterations and the centrality corrector's counts under its switches."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import mbfir
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 16384)
f, a, d = mbfir.spec.spec_h1_dualband(n)
o = mbfir.make_opts(grid_m=m)
mbfir.fir_qp_cvx(n, f, a, d, 120.0, 1e6, opts=o)
for tag, env in (("default", {}), ("no guard", {"MBFIR_CORR_GUARD": "0"}), ("refined corrector solve", {"MBFIR_CORR_PLAIN": "0"}), ("corrector off", {"MBFIR_CORRECTOR": "0"})):
    os.environ.update(env)
    t = time.time(); h, s, i = mbfir.fir_qp_cvx(n, f, a, d, 120.0, 1e6, opts=o, info=True); dt = time.time() - t
    for k in env: os.environ.pop(k)
    print("%-26s %s %3d iterations (%d extended), correctors %d taken %d, %.1f ms = %.3f ms/iteration, pcost %.10e relgap %.1e dres %.1e" % (
        tag, s, i["iters"], i["dd_iters"], i["correctors"], i["correctors_taken"], 1e3 * dt, 1e3 * dt / max(1, i["iters"]), i["pcost"], i["relgap"], i["dres"]), flush=True)
for passes in (2, 3):
    os.environ["MBFIR_CORR_DD_PASSES"] = str(passes)
    t = time.time(); h, s, i = mbfir.fir_qp_cvx(n, f, a, d, 120.0, 1e6, opts=o, info=True); dt = time.time() - t
    os.environ.pop("MBFIR_CORR_DD_PASSES")
    print("%-26s %s %3d iterations (%d extended), correctors %d taken %d, %.1f ms = %.3f ms/iteration, pcost %.10e relgap %.1e dres %.1e" % (
        "corrector: %d passes" % passes, s, i["iters"], i["dd_iters"], i["correctors"], i["correctors_taken"], 1e3 * dt, 1e3 * dt / max(1, i["iters"]), i["pcost"], i["relgap"], i["dres"]), flush=True)
