"""Row-sharded solve of one golden case in loop-back (tests/test_shard_gpu.py's harness), verbose on rank 0."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from conftest import CASES
import test_shard_gpu as T
name = sys.argv[1] if len(sys.argv) > 1 else "ap_c13_58"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 2
fn, args = CASES[name]
h0, s0, i0 = getattr(mbfir, fn)(*args, info=True)
print("unsharded:", s0, i0["iters"], i0["pcost"], i0["relgap"], i0["pres"], i0["dres"], "taken", i0["correctors_taken"], flush=True)
_mk = mbfir.make_opts
def _verbose_opts(**kw):
    if os.environ.get("SHARD_VERBOSE") and kw.get("shard_rank", 0) == 0 and kw.get("shard_size", 1) > 1:
        kw["verbose"] = 1
    return _mk(**kw)
mbfir.make_opts = _verbose_opts
res = T._run_sharded(fn, args, size)
for r in res:
    if isinstance(r, Exception): print("EXC", r); continue
    h, s, i = r
    print("sharded:", s, i["rc"], i["iters"], i["pcost"], i["relgap"], i["pres"], i["dres"], "taken", i["correctors_taken"], "collectives", i["collectives"], flush=True)
