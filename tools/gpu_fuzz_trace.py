"""Per-iteration trace (opts.verbose, stderr) of one fuzz seed on the device."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import mbfir
from test_fuzz_gpu import make_case
which, args = make_case(int(sys.argv[1]))
h, s, i = getattr(mbfir, which)(*args, info=True, opts=mbfir.make_opts(verbose=1))
print(s, i["iters"], i["correctors_taken"], i["correctors"])
