"""One seed of the fuzz campaign in detail: device and oracle certificates at the oracle's stopping iteration and one later."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from oracle import designers
import test_fuzz_gpu as F
seed = int(sys.argv[1])
which, args = F.make_case(seed)
ho, so, io = getattr(designers, which)(*args, info=True)
print("oracle:", so, {k: io[k] for k in ("iters", "pcost", "pres", "dres", "gap", "relgap") if k in io})
hg, sg, ig = getattr(mbfir, which)(*args, info=True)
print("device:", sg, {k: ig[k] for k in ("iters", "pcost", "pres", "dres", "gap", "relgap")}, "taps differ by %.3g" % np.abs(hg - ho).max())
for it in (io["iters"] - 1, io["iters"], io["iters"] + 1):
    h2, s2, i2 = getattr(mbfir, which)(*args, info=True, opts=mbfir.make_opts(max_iter=it))
    print("device, max_iter %d:" % it, s2, {k: i2[k] for k in ("iters", "rc", "pcost", "pres", "dres", "gap", "relgap")})
    h3, s3, i3 = getattr(designers, which)(*args, info=True, max_iter=it)
    print("oracle, max_iter %d:" % it, s3, {k: i3[k] for k in ("iters", "status", "pcost", "pres", "dres", "gap", "relgap") if k in i3},
          "taps device-oracle %.3g" % (np.abs(h2 - h3).max() if len(h2) == len(h3) and len(h2) else -1))
