import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np, mbfir
from conftest import CASES
base = CASES["qp_modelA48"][1]
def widened(f, dfw):
    f = np.asarray(f, float).copy(); f[0::2] -= dfw; f[1::2] += dfw; return list(f)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
jobs = [("fir_qp_cvx", (base[0], widened(base[1], 2e-2 * q), base[2], base[3], base[4], 1e6)) for q in range(4)]
if which == "all":
    jobs += [("fir_qp_cvx", (n, base[1], base[2], base[3], base[4], 1e5)) for n in (40, 44, 52)]
elif which == "obj":
    jobs = [("fir_qp_cvx", (base[0], base[1], base[2], [v * sc for v in base[3]], base[4], obj)) for sc, obj in ((1.0, 1e6), (1.0, 1e8), (0.5, 1e7), (0.3, 1e8), (1.0, 1e4))]
elif which != "edges":
    jobs = [jobs[int(c)] for c in which]
ctx = mbfir.Context(0)
res = mbfir.solve_batch(jobs, ctxs=[ctx], info=True, opts=mbfir.make_opts(lanes=8))
for q, (job, (h, st, i)) in enumerate(zip(jobs, res)):
    if os.environ.get("SINGLE_KP"): os.environ["MBFIR_TEST_CAP_KP"] = os.environ["SINGLE_KP"]
    h1, s1, i1 = mbfir.fir_qp_cvx(*job[1], ctx=ctx, info=True)
    os.environ.pop("MBFIR_TEST_CAP_KP", None)
    same = st == s1 and i["pcost"] == i1["pcost"] and (h is None or np.array_equal(h, h1))
    print(q, "lanes", i["lanes"], st, s1, "iters", i["iters"], i1["iters"], "dd", i["dd_iters"], i1["dd_iters"], "kmax", i["dd_kmax"], i1["dd_kmax"],
          "rows", i["n_rows"], "N", i["n_unknowns"], "pcost %.17g %.17g" % (i["pcost"], i1["pcost"]), "SAME" if same else "DIFFERENT")
