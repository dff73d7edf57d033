"""Dense row-sharded build, its two forms (MBFIR_AR_OVERLAP) against the unsharded solves: taps, objective, conic solution."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from conftest import c13, relinf       # noqa: E402
from test_shard_gpu import _run_sharded
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
fn, args = "fir_ap_cvx", (n,) + tuple(c13(n)) + (0.1, 1e-3)
if len(sys.argv) > 2 and sys.argv[2] == "twoband":
    args = (n, [-0.6, -0.35, -0.1, 0.15, 0.45, 0.8], [0, 0, 0.7, 0.7, 0, 0], [0.01, 0.02, 0.01], 0.1, 1e-2)
def show(tag, h, s, i, z, ref=None):
    print("%-28s %s iters %3d pcost %.12e" % (tag, s, i["iters"], i["pcost"]), end="")
    if ref is not None:
        print("  taps %.2e  z %.2e" % (relinf(h, ref[0]), np.abs(z - ref[1]).max() / np.abs(ref[1]).max()), end="")
    print()
h0, s0, i0 = mbfir.fir_ap_cvx(*args, info=True); z0 = mbfir.get_context().last_solution(i0["n_unknowns"])
show("lattice", h0, s0, i0, z0)
h, s, i = mbfir.fir_ap_cvx(*args, info=True, opts=mbfir.make_opts(dense_trig=1)); z = mbfir.get_context().last_solution(i["n_unknowns"])
show("dense", h, s, i, z, (h0, z0))
for ch in (1, 2, 4):
    os.environ["MBFIR_AR_OVERLAP"] = "2"; os.environ["MBFIR_AR_CHUNKS"] = str(ch)
    h, s, i = mbfir.fir_ap_cvx(*args, info=True, opts=mbfir.make_opts(dense_trig=1)); z = mbfir.get_context().last_solution(i["n_unknowns"])
    show("dense chunked x%d" % ch, h, s, i, z, (h0, z0))
os.environ.pop("MBFIR_AR_CHUNKS")
for mode in ("1", "0"):
    os.environ["MBFIR_AR_OVERLAP"] = mode
    for size in (2, 3):
        res = _run_sharded(fn, args, size, dense=1)
        for r, (h, s, i) in enumerate(res):
            show("sharded/%d overlap=%s rank %d" % (size, mode, r), h, s, i, i["_z"], (h0, z0))
