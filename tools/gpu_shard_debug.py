"""Row-sharded solve of one named case on one GPU with the loop-back all-reduce of tests/test_shard_gpu.py."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import conftest  # noqa: F401
import mbfir
from conftest import CASES
import test_shard_gpu as T
name = sys.argv[1]; size = int(sys.argv[2]) if len(sys.argv) > 2 else 2
fn, args = CASES[name]
h0, s0, i0 = getattr(mbfir, fn)(*args, info=True)
print("unsharded:", s0, i0["iters"], i0["pcost"], i0["relgap"], i0["pres"], i0["dres"])
res = T._run_sharded(fn, args, size)
for r in res:
    if isinstance(r, Exception): print("EXC", r); continue
    h, s, i = r
    print("rank:", s, i["rc"], i["iters"], i["pcost"], i["relgap"], i["pres"], i["dres"], i["n_freq"], i["lattice"])
