"""fir_qprog_phs with the centrality corrector on the orthant rows alone (MBFIR_CORR_BIG=0: round 6's first form) against orthant rows
+ big cone: iterations and time over the fir_qprog_phs specs among tests/test_fuzz_gpu's seeds lo..hi-1, and one min-order search."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from test_fuzz_gpu import make_case
lo, hi = int(sys.argv[1]), int(sys.argv[2])
cases = [c for c in (make_case(s) for s in range(lo, hi)) if c[0] == "fir_qprog_phs"]
for which, args in cases[:4]:
    mbfir.fir_qprog_phs(*args)
res = {}
for tag, env in (("orthant rows + big cone", {}), ("orthant rows alone", {"MBFIR_CORR_BIG": "0"}), ("no corrector", {"MBFIR_CORRECTOR": "0"})):
    os.environ.update(env)
    t = time.time(); its = 0; taken = 0; nc = 0; hs = []
    for which, args in cases:
        h, s, i = mbfir.fir_qprog_phs(*args, info=True)
        its += i["iters"]; taken += i["correctors_taken"]; nc += i["correctors"]; hs.append((s, h))
    dt = time.time() - t
    for k in env: os.environ.pop(k)
    res[tag] = hs
    print("%-26s %d designs: %5d iterations (%.1f per design), correctors taken %d / %d, %.1f ms per design" % (tag, len(cases), its, its / len(cases), taken, nc, 1e3 * dt / len(cases)), flush=True)
a, b = res["orthant rows + big cone"], res["orthant rows alone"]
print("verdicts equal: %s; largest tap difference between the two forms %.2e" % (all(x[0] == y[0] for x, y in zip(a, b)),
      max(np.abs(x[1] - y[1]).max() / max(np.abs(y[1]).max(), 1e-3) for x, y in zip(a, b) if x[0] == y[0] == "Solved")))
