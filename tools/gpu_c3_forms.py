"""BASELINE config 3 as written (fir_qp_cvx, k=120, obj=1e6) through the two forms of the extended-precision KKT solve: the capacitance
form in plain double on the matrix cores (capkkt.hip, the default) and the double-double factorisation (ddlin.hip, MBFIR_DDFORM=dd):
verdict, iterations, objective, taps against each other and against the committed fixture, time alone."""
import json, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
g = json.load(open(os.path.join(ROOT, "tests/golden/h1qp_golden.json")))
cases = [(384, 1536, None), (384, 6144, "h1qp_384_6144"), (512, 16384, "h1qp_512_16384")]
if len(sys.argv) > 1: cases = cases[:int(sys.argv[1])]
for n, m, key in cases:
    f, a, d = mbfir.spec.spec_h1_dualband(n)
    out = {}
    for form in ("cap", "dd"):
        os.environ["MBFIR_DDFORM"] = form
        o = mbfir.make_opts(grid_m=m)
        mbfir.fir_qp_cvx(n, f, a, d, 120.0, 1e6, opts=o)
        t = time.time(); h, s, i = mbfir.fir_qp_cvx(n, f, a, d, 120.0, 1e6, opts=o, info=True); dt = time.time() - t
        out[form] = (h, s, i)
        line = "n=%d m=%d %-3s: %s, %d iterations (%d extended, k max %d), pcost %.12e relgap %.1e pres %.1e dres %.1e, %.3f s" % (
            n, m, form, s, i["iters"], i["dd_iters"], i["dd_kmax"], i["pcost"], i["relgap"], i["pres"], i["dres"], dt)
        if key and s == "Solved":
            hg = np.array(g[key]["h_re"]) + 1j * np.array(g[key]["h_im"])
            line += "; fixture: %d iterations, pcost diff %.1e, taps %.1e" % (g[key]["iters"], abs(i["pcost"] - g[key]["pcost"]) / abs(g[key]["pcost"]), np.abs(h - hg).max() / np.abs(hg).max())
        print(line, flush=True)
    if out["cap"][1] == out["dd"][1] == "Solved":
        print("      cap vs dd: taps %.2e, pcost %.1e" % (np.abs(out["cap"][0] - out["dd"][0]).max() / np.abs(out["dd"][0]).max(), abs(out["cap"][2]["pcost"] - out["dd"][2]["pcost"]) / abs(out["dd"][2]["pcost"])), flush=True)
os.environ.pop("MBFIR_DDFORM", None)
