"""Fuzz seeds (tests/test_fuzz_gpu.make_case): device against oracle with the corrector's counts, under the corrector's switches."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from oracle import designers
from test_fuzz_gpu import make_case
for seed in [int(v) for v in sys.argv[1:]]:
    which, args = make_case(seed)
    ho, so, io = getattr(designers, which)(*args, info=True)
    print("seed %d %s n=%d   oracle: %s %d iterations, correctors %s/%s, pcost %.12e" % (seed, which, args[0], so, io["iters"], io.get("correctors_taken"), io.get("correctors"), io["pcost"]))
    for tag, env in (("default", {}), ("no guard", {"MBFIR_CORR_GUARD": "0"}), ("refined", {"MBFIR_CORR_PLAIN": "0"}), ("off", {"MBFIR_CORRECTOR": "0"})):
        os.environ.update(env)
        hg, sg, ig = getattr(mbfir, which)(*args, info=True)
        for k in env: os.environ.pop(k)
        print("   device %-9s %s %d iterations, correctors %d/%d, pcost %.12e, taps vs oracle %.2e" % (tag, sg, ig["iters"], ig["correctors_taken"], ig["correctors"], ig["pcost"],
              np.abs(hg - ho).max() / max(np.abs(ho).max(), 1e-3) if sg == so == "Solved" else -1))
